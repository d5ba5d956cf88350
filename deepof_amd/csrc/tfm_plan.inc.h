// Transformer family: parameter layout, workspace, weight-gradient jobs and the launch sequences of
// TFMEncoderPT / TFMDecoderPT (models_new.py:985-1327).  Included by vade.hip inside its anonymous namespace, after
// the shared helpers (Carver, JobBuilder, censnet_forward / _backward, head_forward / _backward, run_jobset).
#pragma once

// ---- parameters ---------------------------------------------------------------------------------
// VaDEPT / VQVAEPT / ContrastivePT(encoder_type="transformer").state_dict() order.
void build_tfm_param_layout(DofVadePlan* p) {
  TfmPlan& tf = p->tf;
  const int L = p->L, D = tf.D, DFF = tf.DFF;
  const char* tn[2] = {"encoder.node_tf", "encoder.edge_tf"};
  const int F[2] = {3, 1};
  p->seg_lo[DOF_SEG_ENCODER] = 0;
  for (int s = 0; s < 2; ++s) {
    const std::string pre = tn[s];
    add_shaped(p, pre + ".embed.weight", {D, F[s]}, &tf.emb_w[s]);
    add_shaped(p, pre + ".embed.bias", {D}, &tf.emb_b[s]);
    for (int l = 0; l < 2; ++l) {
      TfmEncLayerOff& o = tf.el[s][l];
      const std::string lp = pre + ".layers." + std::to_string(l);
      int64_t dummy;
      add_shaped(p, lp + ".mha.q_proj.weight", {D, D}, &o.wqkv);  // q | k | v are consecutive: one (3D, D) matrix
      add_shaped(p, lp + ".mha.k_proj.weight", {D, D}, &dummy);
      add_shaped(p, lp + ".mha.v_proj.weight", {D, D}, &dummy);
      add_shaped(p, lp + ".mha.out_proj.weight", {D, D}, &o.wo);
      add_shaped(p, lp + ".norm1.weight", {D}, &o.n1w);
      add_shaped(p, lp + ".norm1.bias", {D}, &o.n1b);
      add_shaped(p, lp + ".ffn.0.weight", {DFF, D}, &o.f0w);
      add_shaped(p, lp + ".ffn.0.bias", {DFF}, &o.f0b);
      add_shaped(p, lp + ".ffn.2.weight", {D, DFF}, &o.f2w);
      add_shaped(p, lp + ".ffn.2.bias", {D}, &o.f2b);
      add_shaped(p, lp + ".norm2.weight", {D}, &o.n2w);
      add_shaped(p, lp + ".norm2.bias", {D}, &o.n2b);
    }
  }
  add_shaped(p, "encoder.spatial_gnn_block.node_kernel", {D, L}, &p->c_nk);
  add_shaped(p, "encoder.spatial_gnn_block.edge_kernel", {D, L}, &p->c_ek);
  add_shaped(p, "encoder.spatial_gnn_block.node_weights", {D, 1}, &p->c_nw);
  add_shaped(p, "encoder.spatial_gnn_block.edge_weights", {D, 1}, &p->c_ew);
  add_shaped(p, "encoder.spatial_gnn_block.node_bias", {L}, &p->c_nb);
  add_shaped(p, "encoder.spatial_gnn_block.edge_bias", {L}, &p->c_eb);
  add_head_params(p);
  p->seg_hi[DOF_SEG_ENCODER] = p->param_total;
  if (p->kind == 2) {
    for (int sg = DOF_SEG_DECODER; sg < DOF_SEG_COUNT; ++sg) p->seg_lo[sg] = p->seg_hi[sg] = p->param_total;
    add_distill_head(p);
    return;
  }
  p->seg_lo[DOF_SEG_DECODER] = p->param_total;
  const int D4 = tf.D4, C3 = p->C3;
  const int le_out[3] = {L, 2 * L, 4 * L}, le_in[3] = {L, L, 2 * L};
  for (int k = 0; k < 3; ++k) {
    const std::string pre = "decoder.latent_expand." + std::to_string(2 * k);
    add_shaped(p, pre + ".weight", {le_out[k], le_in[k]}, &tf.le_w[k]);
    add_shaped(p, pre + ".bias", {le_out[k]}, &tf.le_b[k]);
  }
  for (int l = 0; l < 2; ++l) {
    TfmDecLayerOff& o = tf.dl[l];
    const std::string lp = "decoder.layers." + std::to_string(l);
    int64_t dummy;
    add_shaped(p, lp + ".q_proj.weight", {D4, D4}, &o.wqkv);
    add_shaped(p, lp + ".k_proj.weight", {D4, D4}, &dummy);
    add_shaped(p, lp + ".v_proj.weight", {D4, D4}, &dummy);
    add_shaped(p, lp + ".out_proj.weight", {D4, D4}, &o.wo);
    add_shaped(p, lp + ".norm1.weight", {D4}, &o.n1w);
    add_shaped(p, lp + ".norm1.bias", {D4}, &o.n1b);
    add_shaped(p, lp + ".norm2.weight", {D4}, &o.n2w);
    add_shaped(p, lp + ".norm2.bias", {D4}, &o.n2b);
    add_shaped(p, lp + ".ffn.0.weight", {DFF, D4}, &o.f0w);
    add_shaped(p, lp + ".ffn.0.bias", {DFF}, &o.f0b);
    add_shaped(p, lp + ".ffn.3.weight", {D4, DFF}, &o.f3w);
    add_shaped(p, lp + ".ffn.3.bias", {D4}, &o.f3b);
  }
  add_shaped(p, "decoder.output_proj.weight", {C3, D4}, &tf.out_w);
  add_shaped(p, "decoder.output_proj.bias", {C3}, &tf.out_b);
  add_shaped(p, "decoder.prob_decoder.loc_projection.weight", {C3, C3}, &p->dpw);
  add_shaped(p, "decoder.prob_decoder.loc_projection.bias", {C3}, &p->dpb);
  p->seg_hi[DOF_SEG_DECODER] = p->param_total;
  add_latent_params(p);
}

// dropout sites in the order the reference's forward draws them (oracle/tfm.py names)
void build_tfm_sites(DofVadePlan* p) {
  TfmPlan& tf = p->tf;
  tf.sites.clear();
  int64_t off = 0;
  auto add = [&](const std::string& name, int64_t numel, float prob) {
    tf.sites.push_back({name, numel, off, prob});
    off += numel;
  };
  const int T = p->T;
  const char* sn[2] = {"enc.node", "enc.edge"};
  for (int s = 0; s < 2; ++s) {
    const int64_t S = p->B * (s == 0 ? p->N : p->E);
    add(std::string(sn[s]) + ".embed", S * T * tf.D, 0.1f);
    for (int l = 0; l < 2; ++l) {
      const std::string lp = std::string(sn[s]) + ".l" + std::to_string(l);
      add(lp + ".attn", S * tf.H * T * T, 0.1f);
      add(lp + ".drop1", S * T * tf.D, 0.1f);
      add(lp + ".drop2", S * T * tf.D, 0.1f);
    }
  }
  if (p->kind == 2) return;
  for (int pass = 0; pass < (p->kind == 1 ? 2 : 1); ++pass)
    for (int l = 0; l < 2; ++l) {
      const std::string lp = std::string(pass ? "dec2.l" : "dec.l") + std::to_string(l);
      add(lp + ".attn", p->B * tf.HD * T * T, 0.2f);
      add(lp + ".drop1", p->B * T * tf.D4, 0.2f);
      add(lp + ".ffn", p->B * T * tf.DFF, 0.2f);
      add(lp + ".drop2", p->B * T * tf.D4, 0.2f);
    }
}

DofDrop tfm_drop(const DofVadePlan* p, int site, bool active) {
  const TfmPlan& tf = p->tf;
  DofDrop d;
  memset(&d, 0, sizeof(d));
  if (!active) return d;
  const TfmSite& s = tf.sites[site];
  d.inject = tf.inject ? tf.inject + s.offset : nullptr;
  d.ctr = tf.ext_ctr ? tf.ext_ctr : reinterpret_cast<const uint32_t*>(p->ws + tf.ctr);
  d.seed = (tf.seed ^ 0x9E3779B9u) * (2654435761u + 2u * (uint32_t)site);
  d.thresh = (uint32_t)((double)s.p * 4294967296.0);
  d.scale = 1.0f / (1.0f - s.p);
  return d;
}

// ---- workspace ----------------------------------------------------------------------------------
void build_tfm_workspace_layout(DofVadePlan* p) {
  TfmPlan& tf = p->tf;
  const int L = p->L, T = p->T, D = tf.D, DFF = tf.DFF;
  Carver cv;
  for (int s = 0; s < 2; ++s) {
    StreamWs& w = p->sw[s];
    TfmEncWs& e = tf.ew[s];
    w.G = s == 0 ? p->N : p->E;
    w.F = s == 0 ? 3 : 1;
    w.S = p->B * w.G;
    w.Sp = dof_pad64(w.S);
    const int64_t Sp = w.Sp, rows = (int64_t)T * Sp;
    e.xs = cv.take(rows * w.F);
    e.pad = cv.take(rows);
    e.y0 = cv.take(rows * D);
    for (int l = 0; l < 2; ++l) {
      e.qkv[l] = cv.take(rows * 3 * D); e.ao[l] = cv.take(rows * D);
      e.u1[l] = cv.take(rows * D); e.x1[l] = cv.take(rows * D);
      e.f1[l] = cv.take(rows * DFF);
      e.u2[l] = cv.take(rows * D); e.x2[l] = cv.take(rows * D);
      e.dH1[l] = cv.take(rows * D); e.dH2[l] = cv.take(rows * D);
      e.dF[l] = cv.take(rows * DFF); e.dQKV[l] = cv.take(rows * 3 * D);
    }
    e.tmp = cv.take(rows * D);
    e.dA = cv.take(rows * D); e.dB = cv.take(rows * D); e.dAO = cv.take(rows * D); e.dE = cv.take(rows * D);
    e.ln_blocks = dof_tfm_ln_blocks(D, T, Sp);
    for (int k = 0; k < 4; ++k) e.lnp[k] = cv.take(e.ln_blocks * 2 * D);
    // CensNet operands
    w.n2 = cv.take((int64_t)D * Sp);
    w.dn2 = cv.take((int64_t)D * Sp);
    w.dots = cv.take(Sp);
    w.Y = cv.take((int64_t)D * Sp);
    w.Z = cv.take((int64_t)L * Sp);
    w.dZ = cv.take((int64_t)L * Sp);
    w.dY = cv.take((int64_t)D * Sp);
    w.dd = cv.take(Sp);
    take_triplets(p, cv, s);
  }
  const int64_t Bp = p->Bp;
  p->flat = cv.take((int64_t)p->J * Bp);
  p->enc = cv.take((int64_t)L * Bp);
  p->denc = cv.take((int64_t)L * Bp);
  p->dflat = cv.take((int64_t)p->J * Bp);
  tf.enc_pre = cv.take((int64_t)L * Bp);
  tf.denc_pre = cv.take((int64_t)L * Bp);
  tf.bstat = cv.take(3 * L);
  tf.ctr = cv.take(1);
  tf.pe_enc = cv.take((int64_t)T * D);
  p->lat_blocks = dof_cdiv(p->B, 256);
  p->cl_blocks = dof_cdiv(p->B, CL_ROWS);
  p->cl_zn = cv.take(2 * p->B * L);
  p->cl_inv = cv.take(2 * p->B);
  p->cl_rn = cv.take(2 * p->B);
  p->cl_rowstat = cv.take(4 * p->B);
  p->cl_theta = cv.take(p->B);
  p->cl_partial = cv.take(3 * p->cl_blocks);
  take_head_buffers(p, cv);
  if (p->kind != 2) {
    take_latent_buffers(p, cv);
    TfmDecWs& d = tf.dw;
    const int D4 = tf.D4, C3p = tf.C3p;
    const int64_t rows = (int64_t)T * Bp;
    p->valid = cv.take(rows);
    p->len_d = cv.take(Bp);
    p->dloc = cv.take(64);
    p->dzdec = cv.take(2LL * L * Bp);
    tf.pe_dec = cv.take((int64_t)T * D4);
    d.a1 = cv.take((int64_t)L * Bp); d.g1 = cv.take((int64_t)L * Bp);
    d.a2 = cv.take(2LL * L * Bp); d.g2 = cv.take(2LL * L * Bp);
    d.a3 = cv.take(4LL * L * Bp); d.g3 = cv.take(4LL * L * Bp);
    d.h0 = cv.take(rows * D4);
    for (int l = 0; l < 2; ++l) {
      d.xn1[l] = cv.take(rows * D4); d.qkv[l] = cv.take(rows * 3 * D4); d.ao[l] = cv.take(rows * D4);
      d.hmid[l] = cv.take(rows * D4); d.xn2[l] = cv.take(rows * D4);
      d.fpre[l] = cv.take(rows * DFF); d.f[l] = cv.take(rows * DFF); d.hout[l] = cv.take(rows * D4);
      d.dH1[l] = cv.take(rows * D4); d.dH2[l] = cv.take(rows * D4);
      d.dF[l] = cv.take(rows * DFF); d.dQKV[l] = cv.take(rows * 3 * D4);
    }
    d.tmp = cv.take(rows * D4);
    d.o = cv.take(rows * C3p); d.loc = cv.take(rows * C3p); d.dloc = cv.take(rows * C3p); d.dO = cv.take(rows * C3p);
    d.dR[0] = cv.take(rows * D4); d.dR[1] = cv.take(rows * D4);
    d.dB = cv.take(rows * D4); d.dM = cv.take(rows * D4); d.dAO = cv.take(rows * D4);
    d.dg3 = cv.take(4LL * L * Bp); d.da3 = cv.take(4LL * L * Bp); d.da2 = cv.take(2LL * L * Bp); d.da1 = cv.take((int64_t)L * Bp);
    d.ln_blocks = dof_tfm_ln_blocks(D4, T, Bp);
    for (int k = 0; k < 4; ++k) d.lnp[k] = cv.take(d.ln_blocks * 2 * D4);
  }
  take_tables(p, cv);
}

// ---- weight-gradient jobs -----------------------------------------------------------------------
// dW[o][i] = sum_rows dY[row][o] * X[row][i] for a dense layer (out CO, in CI); dY / X are [r][ldy] / [r][ldx]
// activations.  Bias gradient (row sums of dY) when b_off >= 0.
void tfm_dense_jobs(JobBuilder& jb, const float* dY, int ldy, int CO, const float* X, int ldx, int CI, int64_t w_off,
                    int64_t b_off, int T, int64_t Sp) {
  const int first_job = (int)jb.jobs.size();
  for (int r0 = 0; r0 < CO; r0 += 64) {
    const int rows = CO - r0 < 64 ? CO - r0 : 64;
    int job = -1;
    for (int c0 = 0; c0 < CI; c0 += 16) {
      if (job < 0 || jb.jobs[job].n_tiles == 4) {
        const bool first = job < 0;
        job = jb.add_job(aos(dY, ldy, Sp, r0), rows, T, Sp);
        if (first && b_off >= 0) jb.add_fin(job, 64, rows, 1, rows, rows, b_off + r0, 1, 1);
      }
      const int nc = CI - c0 < 16 ? CI - c0 : 16;
      const int tl = jb.add_tile(job, aos(X, ldx, Sp, c0), nc, 0);
      jb.add_fin(job, tl * 16, rows, nc, rows, rows, w_off + (int64_t)r0 * CI + c0, CI, 1);
    }
  }
  jb.group(first_job, (int)jb.jobs.size() - first_job);   // the layer's row blocks x tile groups share dY and X rows
}

// per-window dense layer ([c][Bp] operands): dW[o][i] = sum_b dpre[o][b] * in[i][b]
void tfm_soa_dense_jobs(JobBuilder& jb, const float* dpre, int CO, const float* in, int CI, int64_t w_off, int64_t b_off,
                        int64_t Bp) {
  for (int c0 = 0; c0 < CI; c0 += 16) {
    const int job = jb.add_job(soa(dpre, Bp), CO, 1, Bp);
    const int nc = CI - c0 < 16 ? CI - c0 : 16;
    jb.add_tile(job, soa(in, Bp, c0), nc, 0);
    jb.add_fin(job, 0, CO, nc, CO, CO, w_off + c0, CI, 1);
    if (c0 == 0) jb.add_fin(job, 64, CO, 1, CO, CO, b_off, 1, 1);
  }
}

void build_tfm_jobs(DofVadePlan* p) {
  TfmPlan& tf = p->tf;
  const int L = p->L, T = p->T, D = tf.D, DFF = tf.DFF;
  float* ws = p->ws;
  const int64_t Bp = p->Bp;
  JobBuilder jb(p->js_enc);
  p->js_enc.wgrads.clear();
  p->js_enc.wg_blocks = 0;
  for (int s = 0; s < 2; ++s) {
    const StreamWs& w = p->sw[s];
    const TfmEncWs& e = tf.ew[s];
    const int64_t Sp = w.Sp;
    tfm_dense_jobs(jb, ws + e.dE, D, D, ws + e.xs, w.F, w.F, tf.emb_w[s], tf.emb_b[s], T, Sp);
    for (int l = 0; l < 2; ++l) {
      const TfmEncLayerOff& o = tf.el[s][l];
      const float* xin = ws + (l == 0 ? e.y0 : e.x2[l - 1]);
      if (l == 0) {
        tfm_dense_jobs(jb, ws + e.dQKV[l], 3 * D, 3 * D, xin, D, D, o.wqkv, -1, T, Sp);
        tfm_dense_jobs(jb, ws + e.dH1[l], D, D, ws + e.ao[l], D, D, o.wo, -1, T, Sp);
        tfm_dense_jobs(jb, ws + e.dF[l], DFF, DFF, ws + e.x1[l], D, D, o.f0w, o.f0b, T, Sp);
        tfm_dense_jobs(jb, ws + e.dH2[l], D, D, ws + e.f1[l], DFF, DFF, o.f2w, o.f2b, T, Sp);
      } else {
        // final layer: only its LAST query row feeds the model output, so everything behind the attention lives in
        // the last time step (rows [(T-1) Sp, T Sp), contiguous); keys / values still come from every step
        const int64_t ro = (int64_t)(T - 1) * Sp;
        tfm_dense_jobs(jb, ws + e.dQKV[l] + ro * 3 * D, 3 * D, D, xin + ro * D, D, D, o.wqkv, -1, 1, Sp);            // W_q
        tfm_dense_jobs(jb, ws + e.dQKV[l] + D, 3 * D, 2 * D, xin, D, D, o.wqkv + (int64_t)D * D, -1, T, Sp);          // W_k | W_v
        tfm_dense_jobs(jb, ws + e.dH1[l] + ro * D, D, D, ws + e.ao[l] + ro * D, D, D, o.wo, -1, 1, Sp);
        tfm_dense_jobs(jb, ws + e.dF[l] + ro * DFF, DFF, DFF, ws + e.x1[l] + ro * D, D, D, o.f0w, o.f0b, 1, Sp);
        tfm_dense_jobs(jb, ws + e.dH2[l] + ro * D, D, D, ws + e.f1[l] + ro * DFF, DFF, DFF, o.f2w, o.f2b, 1, Sp);
      }
    }
    cens_jobs(p, jb, s);
  }
  head_jobs(p, jb);
  jb.close(p->js_enc);
  if (p->kind != 2) {
    const TfmDecWs& d = tf.dw;
    const int D4 = tf.D4, C3 = p->C3, C3p = tf.C3p;
    for (int v = 0; v < 2; ++v) {
      JobBuilder jd(p->js_dec[v]);
      const float* zin = ws + (v == 0 ? p->z : p->enc);
      tfm_soa_dense_jobs(jd, ws + d.da1, L, zin, L, tf.le_w[0], tf.le_b[0], Bp);
      tfm_soa_dense_jobs(jd, ws + d.da2, 2 * L, ws + d.g1, L, tf.le_w[1], tf.le_b[1], Bp);
      tfm_soa_dense_jobs(jd, ws + d.da3, 4 * L, ws + d.g2, 2 * L, tf.le_w[2], tf.le_b[2], Bp);
      for (int l = 0; l < 2; ++l) {
        const TfmDecLayerOff& o = tf.dl[l];
        tfm_dense_jobs(jd, ws + d.dQKV[l], 3 * D4, 3 * D4, ws + d.xn1[l], D4, D4, o.wqkv, -1, T, Bp);
        tfm_dense_jobs(jd, ws + d.dH1[l], D4, D4, ws + d.ao[l], D4, D4, o.wo, -1, T, Bp);
        tfm_dense_jobs(jd, ws + d.dF[l], DFF, DFF, ws + d.xn2[l], D4, D4, o.f0w, o.f0b, T, Bp);
        tfm_dense_jobs(jd, ws + d.dH2[l], D4, D4, ws + d.f[l], DFF, DFF, o.f3w, o.f3b, T, Bp);
      }
      tfm_dense_jobs(jd, ws + d.dO, C3p, C3, ws + d.hout[1], D4, D4, tf.out_w, tf.out_b, T, Bp);
      tfm_dense_jobs(jd, ws + d.dloc, C3p, C3, ws + d.o, C3p, C3, p->dpw, p->dpb, T, Bp);
      jd.close(p->js_dec[v]);
    }
  } else {
    for (int v = 0; v < 2; ++v) {
      JobBuilder jd(p->js_dec[v]);
      jd.close(p->js_dec[v]);
    }
  }
  {
    JobBuilder gb(p->js_gram);
    const float* zsrc = ws + (p->kind == 0 ? p->z : p->enc);
    const int gj = gb.add_job(soa(zsrc, Bp), L, 1, Bp);
    gb.add_tile(gj, soa(zsrc, Bp), L, 0);
    gb.add_fin(gj, 0, L, L, L, L, p->gram, L, 1);
    gb.close(p->js_gram);
  }
}

// ---- launch sequences ---------------------------------------------------------------------------
DofGemm tfm_gemm(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, int K, int N,
                 int T, int64_t S, int64_t Sp) {
  DofGemm g;
  memset(&g, 0, sizeof(g));
  g.X = X; g.ldx = ldx; g.W = W; g.ldw = ldw; g.bias = bias; g.Y = Y; g.ldy = ldy; g.K = K; g.N = N;
  g.T = T; g.S = S; g.Sp = Sp; g.epi = DOF_EPI_NONE;
  return g;
}

int tfm_encoder_forward(DofVadePlan* p, float* params, const float* x, const float* a, bool train, hipStream_t st) {
  TfmPlan& tf = p->tf;
  float* ws = p->ws;
  const int T = p->T, D = tf.D, DFF = tf.DFF;
  const bool keep = train;
  train = train && p->bn_training;
  if (train) TRY(dof_launch_tfm_tick(tf.ext_ctr ? tf.ext_ctr : reinterpret_cast<uint32_t*>(ws + tf.ctr), st));
  (void)keep;
  for (int s = 0; s < 2; ++s) {
    const StreamWs& w = p->sw[s];
    const TfmEncWs& e = tf.ew[s];
    const int64_t S = w.S, Sp = w.Sp;
    const int site0 = 7 * s;
    TRY(dof_launch_tfm_embed(w.F, s == 0 ? x : a, params + tf.emb_w[s], params + tf.emb_b[s], ws + tf.pe_enc, ws + e.xs,
                             ws + e.pad, ws + e.y0, tfm_drop(p, site0, train), T, w.G, D, S, Sp, st));
    for (int l = 0; l < 2; ++l) {
      const TfmEncLayerOff& o = tf.el[s][l];
      const float* xin = ws + (l == 0 ? e.y0 : e.x2[l - 1]);
      const int sl = site0 + 1 + 3 * l;
      // final layer: only the LAST query row reaches the output (TransformerCorePT returns y[:, -1]): keys and
      // values for every step, everything else on the last time step's rows [ro, ro + Sp) with a one-step launch
      const bool last = l == 1;
      const int Tl = last ? 1 : T;
      const int64_t ro = last ? (int64_t)(T - 1) * Sp : 0;
      if (last) {
        TRY(dof_launch_tfm_gemm(tfm_gemm(xin, D, params + o.wqkv + (int64_t)D * D, D, nullptr, ws + e.qkv[l] + D, 3 * D, D,
                                         2 * D, T, S, Sp), st));
        TRY(dof_launch_tfm_gemm(tfm_gemm(xin + ro * D, D, params + o.wqkv, D, nullptr, ws + e.qkv[l] + ro * 3 * D, 3 * D, D, D,
                                         1, S, Sp), st));
      } else {
        TRY(dof_launch_tfm_gemm(tfm_gemm(xin, D, params + o.wqkv, D, nullptr, ws + e.qkv[l], 3 * D, D, 3 * D, T, S, Sp), st));
      }
      DofAttn at;
      memset(&at, 0, sizeof(at));
      at.qkv = ws + e.qkv[l]; at.ao = ws + e.ao[l]; at.pad = ws + e.pad; at.drop = tfm_drop(p, sl, train);
      at.T = T; at.D = D; at.H = tf.H; at.causal = 0; at.S = S; at.Sp = Sp; at.q_last = last ? 1 : 0;
      TRY(dof_launch_tfm_attn(at, 0, st));
      TRY(dof_launch_tfm_gemm(tfm_gemm(ws + e.ao[l] + ro * D, D, params + o.wo, D, nullptr, ws + e.tmp + ro * D, D, D, D, Tl, S,
                                       Sp), st));
      DofLn ln;
      memset(&ln, 0, sizeof(ln));
      ln.x = xin + ro * D; ln.h = ws + e.tmp + ro * D; ln.u = ws + e.u1[l] + ro * D; ln.y = ws + e.x1[l] + ro * D;
      ln.gamma = params + o.n1w; ln.beta = params + o.n1b; ln.drop = tfm_drop(p, sl + 1, train); ln.T = Tl; ln.S = S;
      ln.Sp = Sp; ln.eps = 1e-6f; ln.t_off = last ? T - 1 : 0; ln.T_idx = T;
      TRY(dof_launch_tfm_add_ln(ln, D, st));
      DofGemm g1 = tfm_gemm(ws + e.x1[l] + ro * D, D, params + o.f0w, D, params + o.f0b, ws + e.f1[l] + ro * DFF, DFF, D, DFF,
                            Tl, S, Sp);
      g1.epi = DOF_EPI_RELU;
      TRY(dof_launch_tfm_gemm(g1, st));
      TRY(dof_launch_tfm_gemm(tfm_gemm(ws + e.f1[l] + ro * DFF, DFF, params + o.f2w, DFF, params + o.f2b, ws + e.tmp + ro * D, D,
                                       DFF, D, Tl, S, Sp), st));
      ln.x = ws + e.x1[l] + ro * D; ln.u = ws + e.u2[l] + ro * D; ln.y = ws + e.x2[l] + ro * D; ln.gamma = params + o.n2w;
      ln.beta = params + o.n2b; ln.drop = tfm_drop(p, sl + 2, train);
      TRY(dof_launch_tfm_add_ln(ln, D, st));
    }
    TRY(dof_launch_tfm_last(ws + e.x2[1], ws + w.n2, T, D, S, Sp, st));
  }
  TRY(censnet_forward(p, params, st));
  TRY(head_forward(p, params, train, ws + tf.enc_pre, st));
  return dof_launch_tfm_bstd(ws + tf.enc_pre, ws + p->enc, ws + tf.bstat, p->L, train && p->B > 1 ? 1 : 0, p->B, p->Bp, st);
}

int tfm_encoder_backward(DofVadePlan* p, const float* params, float* grads, hipStream_t st, int accumulate) {
  TfmPlan& tf = p->tf;
  float* ws = p->ws;
  const int T = p->T, D = tf.D, DFF = tf.DFF;
  const bool train = p->bn_training;
  TRY(dof_launch_tfm_bstd_bwd(ws + p->denc, ws + p->enc, ws + tf.bstat, ws + tf.denc_pre, p->L,
                              train && p->B > 1 ? 1 : 0, p->B, p->Bp, st));
  TRY(head_backward(p, params, grads, accumulate, ws + tf.denc_pre, st));
  TRY(censnet_backward(p, params, st));
  for (int s = 0; s < 2; ++s) {
    const StreamWs& w = p->sw[s];
    const TfmEncWs& e = tf.ew[s];
    const int64_t S = w.S, Sp = w.Sp;
    const int site0 = 7 * s;
    for (int l = 1; l >= 0; --l) {
      const TfmEncLayerOff& o = tf.el[s][l];
      const int sl = site0 + 1 + 3 * l;
      const bool last = l == 1;  // see tfm_encoder_forward: the final layer lives in the last time step
      const int Tl = last ? 1 : T;
      const int64_t ro = last ? (int64_t)(T - 1) * Sp : 0;
      const int t_off = last ? T - 1 : 0;
      // d x2[l]: the CensNet gradient of the last step (final layer) or the layer above's d xin (in dA)
      if (last) TRY(dof_launch_tfm_last_bwd(ws + w.dn2, ws + e.dA + ro * D, 1, D, S, Sp, st));
      // LayerNorm2(u2 = x1 + drop2(ffn)): dA = d x2[l]  ->  dB = d u2 (residual part of d x1), dH2 = d ffn output
      DofLnBwd lb;
      memset(&lb, 0, sizeof(lb));
      lb.dy1 = ws + e.dA + ro * D; lb.u = ws + e.u2[l] + ro * D; lb.gamma = params + o.n2w; lb.du = ws + e.dB + ro * D;
      lb.dh = ws + e.dH2[l] + ro * D; lb.partial = ws + e.lnp[2 * l + 1]; lb.drop = tfm_drop(p, sl + 2, train); lb.T = Tl;
      lb.S = S; lb.Sp = Sp; lb.eps = 1e-6f; lb.t_off = t_off; lb.T_idx = T;
      TRY(dof_launch_tfm_ln_bwd(lb, D, st));
      DofGemm g2 = tfm_gemm(ws + e.dH2[l] + ro * D, D, params + o.f2w, DFF, nullptr, ws + e.dF[l] + ro * DFF, DFF, D, DFF, Tl, S, Sp);
      g2.trans = 1; g2.epi = DOF_EPI_MUL_RELU; g2.aux = ws + e.f1[l] + ro * DFF; g2.ldaux = DFF;
      TRY(dof_launch_tfm_gemm(g2, st));
      DofGemm g1 = tfm_gemm(ws + e.dF[l] + ro * DFF, DFF, params + o.f0w, D, nullptr, ws + e.dA + ro * D, D, DFF, D, Tl, S, Sp);
      g1.trans = 1;
      TRY(dof_launch_tfm_gemm(g1, st));  // dA = d x1 through the ffn
      // LayerNorm1(u1 = xin + drop1(attn out)): dy = dA + dB  ->  d u1 (residual part of d xin), dH1
      float* du1 = last ? ws + e.tmp + ro * D : ws + e.dE;
      lb.dy1 = ws + e.dA + ro * D; lb.dy2 = ws + e.dB + ro * D; lb.u = ws + e.u1[l] + ro * D; lb.gamma = params + o.n1w;
      lb.du = du1; lb.dh = ws + e.dH1[l] + ro * D; lb.partial = ws + e.lnp[2 * l]; lb.drop = tfm_drop(p, sl + 1, train);
      TRY(dof_launch_tfm_ln_bwd(lb, D, st));
      DofGemm go = tfm_gemm(ws + e.dH1[l] + ro * D, D, params + o.wo, D, nullptr, ws + e.dAO + ro * D, D, D, D, Tl, S, Sp);
      go.trans = 1;
      TRY(dof_launch_tfm_gemm(go, st));
      DofAttn at;
      memset(&at, 0, sizeof(at));
      at.qkv = ws + e.qkv[l]; at.dao = ws + e.dAO; at.dqkv = ws + e.dQKV[l]; at.pad = ws + e.pad;
      at.drop = tfm_drop(p, sl, train); at.T = T; at.D = D; at.H = tf.H; at.causal = 0; at.S = S; at.Sp = Sp;
      at.q_last = last ? 1 : 0;
      TRY(dof_launch_tfm_attn(at, 1, st));
      if (last) {
        // d xin = (dK | dV) (W_k | W_v) on every row, + dQ W_q + d u1 on the last step's rows
        DofGemm gk = tfm_gemm(ws + e.dQKV[l] + D, 3 * D, params + o.wqkv + (int64_t)D * D, D, nullptr, ws + e.dE, D, 2 * D, D, T, S, Sp);
        gk.trans = 1;
        TRY(dof_launch_tfm_gemm(gk, st));
        DofGemm gq = tfm_gemm(ws + e.dQKV[l] + ro * 3 * D, 3 * D, params + o.wqkv, D, nullptr, ws + e.dE + ro * D, D, D, D, 1, S, Sp);
        gq.trans = 1; gq.accumulate = 1;
        TRY(dof_launch_tfm_gemm(gq, st));
        DofLn ad;
        memset(&ad, 0, sizeof(ad));
        ad.x = ws + e.dE + ro * D; ad.h = du1; ad.u = ws + e.dE + ro * D; ad.T = 1; ad.S = S; ad.Sp = Sp;
        TRY(dof_launch_tfm_add_ln(ad, D, st));
        // hand over: the layer below reads its incoming gradient from dA
        DofLn cp;
        memset(&cp, 0, sizeof(cp));
        cp.x = ws + e.dE; cp.u = ws + e.dA; cp.T = T; cp.S = S; cp.Sp = Sp;
        TRY(dof_launch_tfm_add_ln(cp, D, st));
      } else {
        // d xin = dE (residual) + dQKV Wqkv  -> accumulated into dE = d y0
        DofGemm gq = tfm_gemm(ws + e.dQKV[l], 3 * D, params + o.wqkv, D, nullptr, ws + e.dE, D, 3 * D, D, T, S, Sp);
        gq.trans = 1; gq.accumulate = 1;
        TRY(dof_launch_tfm_gemm(gq, st));
      }
    }
    // embedding: dE = d y0 -> gradient of the pre-activation (in place)
    TRY(dof_launch_tfm_embed_bwd(w.F, ws + e.xs, params + tf.emb_w[s], params + tf.emb_b[s], ws + e.dE, ws + e.dE,
                                 tfm_drop(p, site0, train), T, D, S, Sp, st));
  }
  for (int s = 0; s < 2; ++s) {  // LayerNorm weight / bias gradients: (g * xhat | g) partial sums
    const TfmEncWs& e = tf.ew[s];
    DofSumJobs sj;
    sj.n = 4;
    for (int l = 0; l < 2; ++l) {
      const int64_t nb = l == 1 ? dof_tfm_ln_blocks(D, 1, p->sw[s].Sp) : e.ln_blocks;  // final layer: one time step
      sj.partial[2 * l] = ws + e.lnp[2 * l]; sj.nblk[2 * l] = nb; sj.nv[2 * l] = 2 * D; sj.out[2 * l] = grads + tf.el[s][l].n1w;
      sj.partial[2 * l + 1] = ws + e.lnp[2 * l + 1]; sj.nblk[2 * l + 1] = nb; sj.nv[2 * l + 1] = 2 * D;
      sj.out[2 * l + 1] = grads + tf.el[s][l].n2w;
    }
    TRY(dof_launch_sum_partials_multi(sj, accumulate, st));
  }
  (void)DFF;
  return run_jobset(p, p->js_enc, grads, accumulate, st);
}

// TFMDecoderPT.forward from the latent batch zin [L][Bp]; `second`: the VQ-VAE's pass on the raw encoder output
int tfm_decoder_forward(DofVadePlan* p, const float* params, const float* x, const float* zin, float* recon_partial,
                        bool train, float* loc_out, bool second, hipStream_t st) {
  TfmPlan& tf = p->tf;
  float* ws = p->ws;
  const TfmDecWs& d = tf.dw;
  const int L = p->L, T = p->T, D4 = tf.D4, DFF = tf.DFF, C3 = p->C3, C3p = tf.C3p;
  const int64_t B = p->B, Bp = p->Bp;
  const bool keep = train;
  train = train && p->bn_training;
  tf.dec_second = second;
  const int site0 = 14 + (second ? 8 : 0);
  DOF_LAUNCH(k_dec_valid, (dof_cdiv(B * T, 256)), (256), st, x, T, C3, B, Bp, ws + p->valid);
  TRY(dof_check_launch("k_dec_valid"));
  DofDecExp ex;
  memset(&ex, 0, sizeof(ex));
  ex.z = zin; ex.w0 = params + tf.le_w[0]; ex.b0 = params + tf.le_b[0]; ex.w1 = params + tf.le_w[1];
  ex.b1 = params + tf.le_b[1]; ex.w2 = params + tf.le_w[2]; ex.b2 = params + tf.le_b[2];
  ex.a1 = ws + d.a1; ex.g1 = ws + d.g1; ex.a2 = ws + d.a2; ex.g2 = ws + d.g2; ex.a3 = ws + d.a3; ex.g3 = ws + d.g3;
  ex.keep = keep ? 1 : 0; ex.B = B; ex.Bp = Bp;
  TRY(dof_launch_tfm_dec_expand(L, ex, st));
  TRY(dof_launch_tfm_dec_h0(ws + d.g3, ws + tf.pe_dec, ws + d.h0, T, D4, B, Bp, st));
  for (int l = 0; l < 2; ++l) {
    const TfmDecLayerOff& o = tf.dl[l];
    const float* hin = ws + (l == 0 ? d.h0 : d.hout[l - 1]);
    const int sl = site0 + 4 * l;
    DofLn ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = hin; ln.y = ws + d.xn1[l]; ln.gamma = params + o.n1w; ln.beta = params + o.n1b; ln.T = T; ln.S = B; ln.Sp = Bp;
    ln.eps = 1e-6f;
    TRY(dof_launch_tfm_add_ln(ln, D4, st));
    TRY(dof_launch_tfm_gemm(tfm_gemm(ws + d.xn1[l], D4, params + o.wqkv, D4, nullptr, ws + d.qkv[l], 3 * D4, D4, 3 * D4, T, B, Bp), st));
    DofAttn at;
    memset(&at, 0, sizeof(at));
    at.qkv = ws + d.qkv[l]; at.ao = ws + d.ao[l]; at.drop = tfm_drop(p, sl, train); at.T = T; at.D = D4; at.H = tf.HD;
    at.causal = 1; at.S = B; at.Sp = Bp;
    TRY(dof_launch_tfm_attn(at, 0, st));
    TRY(dof_launch_tfm_gemm(tfm_gemm(ws + d.ao[l], D4, params + o.wo, D4, nullptr, ws + d.tmp, D4, D4, D4, T, B, Bp), st));
    ln.h = ws + d.tmp; ln.u = ws + d.hmid[l]; ln.y = ws + d.xn2[l]; ln.gamma = params + o.n2w; ln.beta = params + o.n2b;
    ln.drop = tfm_drop(p, sl + 1, train);
    TRY(dof_launch_tfm_add_ln(ln, D4, st));
    DofGemm g1 = tfm_gemm(ws + d.xn2[l], D4, params + o.f0w, D4, params + o.f0b, ws + d.f[l], DFF, D4, DFF, T, B, Bp);
    g1.epi = DOF_EPI_GELU; g1.aux_out = ws + d.fpre[l]; g1.drop = tfm_drop(p, sl + 2, train); g1.drop_ld = DFF;
    TRY(dof_launch_tfm_gemm(g1, st));
    TRY(dof_launch_tfm_gemm(tfm_gemm(ws + d.f[l], DFF, params + o.f3w, DFF, params + o.f3b, ws + d.tmp, D4, DFF, D4, T, B, Bp), st));
    DofLn ad;
    memset(&ad, 0, sizeof(ad));
    ad.x = ws + d.hmid[l]; ad.h = ws + d.tmp; ad.u = ws + d.hout[l]; ad.drop = tfm_drop(p, sl + 3, train);
    ad.T = T; ad.S = B; ad.Sp = Bp;
    TRY(dof_launch_tfm_add_ln(ad, D4, st));
  }
  TRY(dof_launch_tfm_gemm(tfm_gemm(ws + d.hout[1], D4, params + tf.out_w, D4, params + tf.out_b, ws + d.o, C3p, D4, C3, T, B, Bp), st));
  TRY(dof_launch_tfm_gemm(tfm_gemm(ws + d.o, C3p, params + p->dpw, C3, params + p->dpb, ws + d.loc, C3p, C3, C3, T, B, Bp), st));
  return dof_launch_tfm_dec_logp(ws + d.loc, C3p, x, ws + p->valid, loc_out, recon_partial, ws + d.dloc, T, C3, keep ? 1 : 0,
                                 B, Bp, st);
}

// Backward of tfm_decoder_forward(train): parameter gradients (set / accumulated), d loss / d zin into slab 0 of
// ws.dzdec (slab 1 stays zero: it is the second GRU direction of the recurrent decoder).
int tfm_decoder_backward(DofVadePlan* p, const float* params, int which_input, float* grads, int accumulate,
                         hipStream_t st) {
  TfmPlan& tf = p->tf;
  float* ws = p->ws;
  const TfmDecWs& d = tf.dw;
  const int L = p->L, T = p->T, D4 = tf.D4, DFF = tf.DFF, C3 = p->C3, C3p = tf.C3p;
  const int64_t B = p->B, Bp = p->Bp;
  const bool train = p->bn_training;
  const int site0 = 14 + (tf.dec_second ? 8 : 0);
  // loc = o Wl^T + bl ; o = h Wo^T + bo
  DofGemm gl = tfm_gemm(ws + d.dloc, C3p, params + p->dpw, C3, nullptr, ws + d.dO, C3p, C3, C3, T, B, Bp);
  gl.trans = 1;
  TRY(dof_launch_tfm_gemm(gl, st));
  DofGemm go = tfm_gemm(ws + d.dO, C3p, params + tf.out_w, D4, nullptr, ws + d.dR[1], D4, C3, D4, T, B, Bp);
  go.trans = 1;
  TRY(dof_launch_tfm_gemm(go, st));  // dR[1] = d hout[1]
  for (int l = 1; l >= 0; --l) {
    const TfmDecLayerOff& o = tf.dl[l];
    const float* hin = ws + (l == 0 ? d.h0 : d.hout[l - 1]);
    const int sl = site0 + 4 * l;
    float* dhout = ws + d.dR[l];
    // hout = hmid + drop2(ffn out): d ffn-out = dhout * keep
    DofLnBwd ab;
    memset(&ab, 0, sizeof(ab));
    ab.dres = dhout; ab.dh = ws + d.dH2[l]; ab.drop = tfm_drop(p, sl + 3, train); ab.T = T; ab.S = B; ab.Sp = Bp;
    TRY(dof_launch_tfm_ln_bwd(ab, D4, st));
    DofGemm g2 = tfm_gemm(ws + d.dH2[l], D4, params + o.f3w, DFF, nullptr, ws + d.dF[l], DFF, D4, DFF, T, B, Bp);
    g2.trans = 1; g2.epi = DOF_EPI_MUL_DGELU; g2.aux = ws + d.fpre[l]; g2.ldaux = DFF; g2.drop = tfm_drop(p, sl + 2, train);
    g2.drop_ld = DFF;
    TRY(dof_launch_tfm_gemm(g2, st));
    DofGemm g1 = tfm_gemm(ws + d.dF[l], DFF, params + o.f0w, D4, nullptr, ws + d.dB, D4, DFF, D4, T, B, Bp);
    g1.trans = 1;
    TRY(dof_launch_tfm_gemm(g1, st));  // dB = d xn2
    // xn2 = LN2(hmid), hmid = hin + drop1(attn out): d hmid = LN2'(dB) + dhout -> dM ; dH1 = dM * keep
    DofLnBwd lb;
    memset(&lb, 0, sizeof(lb));
    lb.dy1 = ws + d.dB; lb.dres = dhout; lb.u = ws + d.hmid[l]; lb.gamma = params + o.n2w; lb.du = ws + d.dM;
    lb.dh = ws + d.dH1[l]; lb.partial = ws + d.lnp[2 * l + 1]; lb.drop = tfm_drop(p, sl + 1, train); lb.T = T; lb.S = B;
    lb.Sp = Bp; lb.eps = 1e-6f;
    TRY(dof_launch_tfm_ln_bwd(lb, D4, st));
    DofGemm gw = tfm_gemm(ws + d.dH1[l], D4, params + o.wo, D4, nullptr, ws + d.dAO, D4, D4, D4, T, B, Bp);
    gw.trans = 1;
    TRY(dof_launch_tfm_gemm(gw, st));
    DofAttn at;
    memset(&at, 0, sizeof(at));
    at.qkv = ws + d.qkv[l]; at.dao = ws + d.dAO; at.dqkv = ws + d.dQKV[l]; at.drop = tfm_drop(p, sl, train);
    at.T = T; at.D = D4; at.H = tf.HD; at.causal = 1; at.S = B; at.Sp = Bp;
    TRY(dof_launch_tfm_attn(at, 1, st));
    DofGemm gq = tfm_gemm(ws + d.dQKV[l], 3 * D4, params + o.wqkv, D4, nullptr, ws + d.dB, D4, 3 * D4, D4, T, B, Bp);
    gq.trans = 1;
    TRY(dof_launch_tfm_gemm(gq, st));  // dB = d xn1
    // xn1 = LN1(hin): d hin = LN1'(dB) + dM
    DofLnBwd l1;
    memset(&l1, 0, sizeof(l1));
    l1.dy1 = ws + d.dB; l1.dres = ws + d.dM; l1.u = hin; l1.gamma = params + o.n1w; l1.du = ws + d.dR[l == 1 ? 0 : 1];
    l1.partial = ws + d.lnp[2 * l]; l1.T = T; l1.S = B; l1.Sp = Bp; l1.eps = 1e-6f;
    TRY(dof_launch_tfm_ln_bwd(l1, D4, st));
  }
  // d h0 is in dR[1]: sum over time -> d g3 -> latent-expand MLP backward -> d zin
  TRY(dof_launch_tfm_dec_sum_time(ws + d.dR[1], ws + d.dg3, T, D4, B, Bp, st));
  DofDecExp ex;
  memset(&ex, 0, sizeof(ex));
  ex.w0 = params + tf.le_w[0]; ex.w1 = params + tf.le_w[1]; ex.w2 = params + tf.le_w[2];
  ex.a1 = ws + d.a1; ex.a2 = ws + d.a2; ex.a3 = ws + d.a3; ex.B = B; ex.Bp = Bp;
  TRY(dof_launch_tfm_dec_expand_bwd(L, ex, ws + d.dg3, ws + d.da3, ws + d.da2, ws + d.da1, ws + p->dzdec, st));
  {
    DofSumJobs sj;
    sj.n = 4;
    for (int l = 0; l < 2; ++l) {
      sj.partial[2 * l] = ws + d.lnp[2 * l]; sj.nblk[2 * l] = d.ln_blocks; sj.nv[2 * l] = 2 * D4; sj.out[2 * l] = grads + tf.dl[l].n1w;
      sj.partial[2 * l + 1] = ws + d.lnp[2 * l + 1]; sj.nblk[2 * l + 1] = d.ln_blocks; sj.nv[2 * l + 1] = 2 * D4;
      sj.out[2 * l + 1] = grads + tf.dl[l].n2w;
    }
    TRY(dof_launch_sum_partials_multi(sj, accumulate, st));
  }
  return run_jobset(p, p->js_dec[which_input], grads, accumulate, st);
}
