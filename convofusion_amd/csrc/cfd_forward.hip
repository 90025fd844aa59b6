// libcfdenoise: the launches of one denoiser forward (tile kernels: enqueue_rows; small problems: enqueue_rows_rt), cfd_forward and the
// per-class profiling hook.
#include "cfd_internal.hpp"

// ---- the denoiser forward: Denoiser.forward (denoiser.py:173-386) --------------------------------------
// Input: c->w->sample_sp (SP [M][128]); time tables built; output: c->w->eps (fp32 [M][128]).


int enqueue_denoise(Ctx* c, hipStream_t st) {
  CHK(enqueue_memside(c, st));
  if (c->w->pb.rt) return enqueue_rows_rt(c, st);
  return enqueue_rows(c, st, 0, c->w->pb.Be);
}

// The forward for small problems: launches of 16-token x 16-feature workgroups (rowtile.hpp); same buffers, same tap points.
// With `sv` (the WEG evaluation, weg_rt.hpp) every residual update goes to a buffer of its own, the self-attention operands, the
// cross-attention scores and the FFN pre-activations of every layer are kept, and the pass ends behind the last layer's
// cross-attention (nothing above it reaches the objective).

int enqueue_rows_rt(Ctx* c, hipStream_t st, const RtSave* sv) {
  const Problem& p = c->w->pb;
  const int nl = c->nl, L = p.L, tpr = (L + 15) / 16, ntile = p.Be * tpr;
  const int* dstep = c->w->d_step.as<int>();
  const int lds_xpv = (p.Sp_tot / 32) * 2048 + 8 * (p.Sp_tot <= 512 ? 2 : 4) * 2048 + 16 * 32 * 16 + 512 + 2048;
#define RT_SET_LDS(kernel, bytes) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes))
  static unsigned long long attr = 0;
  if (!((attr >> (c->cfg.device & 63)) & 1ull)) {
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_EMBED, 256, CFD_LAT / 32, 1>), rt_gemm_lds(RT_PRO_SP, 256, CFD_LAT / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_LN, RT_EPI_QKV, 512, CFD_D / 32, 3>), rt_gemm_lds(RT_PRO_LN, 512, CFD_D / 32, 3));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 1>), rt_gemm_lds(RT_PRO_SP, 256, CFD_D / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 1>), rt_gemm_lds(RT_PRO_ADALN, 512, CFD_D / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_LN, RT_EPI_SPLIT, 512, CFD_D / 32, 2>), rt_gemm_lds(RT_PRO_LN, 512, CFD_D / 32, 2));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_RESID, 512, CFD_FF / 32, 1>), rt_gemm_lds(RT_PRO_SP, 512, CFD_FF / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_LN, RT_EPI_F32, 512, CFD_D / 32, 1>), rt_gemm_lds(RT_PRO_LN, 512, CFD_D / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 2>), rt_gemm_lds(RT_PRO_SP, 256, CFD_D / 32, 2));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 2>), rt_gemm_lds(RT_PRO_ADALN, 512, CFD_D / 32, 2));
    RT_SET_LDS(rt_xscore_kernel<>, RT_XS_LDS);
    RT_SET_LDS(rt_xpv_kernel<512>, (512 / 32) * 2048 + 8 * 2 * 2048 + 16 * 32 * 16 + 512 + 2048);
    RT_SET_LDS(rt_xpv_kernel<RT_MAX_KEYS>, (RT_MAX_KEYS / 32) * 2048 + 8 * 4 * 2048 + 16 * 32 * 16 + 512 + 2048);
    attr |= 1ull << (c->cfg.device & 63);
  }
#undef RT_SET_LDS
  // The residual stream alternates between two buffers: a time block's workgroups read COMPLETE rows (LayerNorm prologue) while the
  // other workgroups of the tile write their 16 features of the sum, so it must not run in place.  x -> (time block 1) -> h ->
  // (cross-attention) -> x -> (time block 2) -> h -> (FFN) -> x; the products whose prologue reads another matrix (out-projection,
  // FFN2) and the cross-attention's second half touch only their own 16 features of the rows and may update in place.
  float* const xw = c->w->x.as<float>();
  float* const hw = c->w->h_sp.as<float>();   // (the tile-kernel path's LayerNorm output: same bytes, unused here)
  auto X = [&](int l, int k) -> float* { return sv ? sv->x[l][k] : ((k == 2 || k == 4) ? hw : xw); };
  // This step's rows of the per-step tables.  One table row (cfd_forward, the WEG evaluation): the tables themselves.  A sampling
  // run: fixed buffers refreshed by ONE launch at the start of the iteration, so that no launch of the iteration has the step index
  // as a dependent scalar load in front of its operand loads.
  const float* ss_now = c->w->ss_tab.as<float>();
  const float *kb_now[CFD_NMEM], *vb_now[CFD_NMEM], *cbt_now[CFD_NMEM];
  for (int j = 0; j < CFD_NMEM; ++j) { kb_now[j] = c->w->kbtab[j].as<float>(); vb_now[j] = c->w->vbtab[j].as<float>(); cbt_now[j] = c->w->rt_cbt[j].as<float>(); }
  if (p.T > 1) {
    RtStepRowsArgs ra;
    memset(&ra, 0, sizeof(ra));
    size_t off4 = 0;
    int nt = 0, nwg = 0;
    auto add = [&](const float*& now, int nfloat) {
      float* dst = c->w->rt_cur.as<float>() + off4 * 4;
      ra.src[nt] = now; ra.dst[nt] = dst; ra.n4[nt] = nfloat / 4; ra.first[nt] = nwg;
      nwg += (nfloat / 4 + 255) / 256; off4 += (size_t)(nfloat / 4 + 63) / 64 * 64; ++nt;
      now = dst;
    };
    size_t need4 = (size_t)(nl * 4 * CFD_D / 4 + 64);
    for (int j = 0; j < CFD_NMEM; ++j) need4 += (size_t)((nl * CFD_D + 32) / 4 + 64) + (size_t)(nl * CFD_D / 4 + 64) + (size_t)((nl + 1) * p.U[j] * p.Sp[j] / 4 + 64);
    CHK(c->w->rt_cur.ensure(need4 * 16));
    add(ss_now, nl * 4 * CFD_D);
    for (int j = 0; j < CFD_NMEM; ++j) { add(kb_now[j], nl * CFD_D + 32); add(vb_now[j], nl * CFD_D); add(cbt_now[j], (nl + 1) * p.U[j] * p.Sp[j]); }
    ra.ntab = nt; ra.first[nt] = nwg; ra.d_step = dstep;
    LAUNCH(CFD_PROF_OTHER, rt_step_rows_kernel<>, dim3(nwg), dim3(256), st, ra);
  }
  c->w->now_ss = ss_now;
  for (int j = 0; j < CFD_NMEM; ++j) { c->w->now_kb[j] = kb_now[j]; c->w->now_vb[j] = vb_now[j]; }
  RtGemmArgs base;
  memset(&base, 0, sizeof(base));
  base.L = L; base.tpr = tpr;
  // Two 16-feature blocks per workgroup for the 512 x 512 residual products from two utterances on: half the workgroups, each normalising
  // its 16 rows once for two blocks (one utterance: 0.411 -> 0.421 s per 1000 steps, two: 0.544 -> 0.528, four: 0.882 -> 0.869; same sums
  // in the same order, so bit-identical).  CFD_RT_NFB2_TILES=<token tiles> moves the threshold (read at cfd_create).
  const bool nfb2 = ntile >= c->rt_nfb2_tiles && !sv;
#define RT_LAUNCH(cls, PRO, EPI, NT, KT, NFB, nfeat, args)                                                        \
  do {                                                                                                          \
    Bracket _br(c, cls, st);                                                                                    \
    hipLaunchKernelGGL((rt_gemm_kernel<PRO, EPI, NT, KT, NFB>), dim3((nfeat) / (16 * NFB), ntile), dim3(NT), rt_gemm_lds(PRO, NT, KT, NFB), st, args); \
    hipError_t _e = hipGetLastError();                                                                          \
    if (_e != hipSuccess) return fail(CFD_E_HIP, "row-tile launch failed: %s", hipGetErrorString(_e));          \
  } while (0)
  // 1. latent embedding + body/hand embedding + query PE          (denoiser.py:187,316-326)
  {
    RtGemmArgs a = base;
    a.a_sp = c->w->sample_sp.as<char>(); a.w = c->we_sp.as<char>(); a.bias = rawp(c, "latent_embd.bias");
    a.bh = rawp(c, "bh_embedding.weight"); a.qpe = rawp(c, "query_pos.pe"); a.xo = X(0, 0);
    RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_EMBED, 256, CFD_LAT / 32, 1, CFD_D, a);
  }
  if (c->stop_stage == 1) return CFD_OK;
  auto time_block = [&](const DBuf& w, const float* g, const float* b, const float* bias, int tbidx, const float* xin, float* xout) -> int {
    RtGemmArgs a = base;
    a.x = xin; a.xr = xin; a.xo = xout;
    a.g = g; a.b = b; a.ss = ss_now + (size_t)tbidx * 2 * CFD_D; a.w = w.as<char>(); a.bias = bias;
    if (nfb2) RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 2, CFD_D, a);
    else RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 1, CFD_D, a);
    return CFD_OK;
  };
  RtXArgs xa;
  memset(&xa, 0, sizeof(xa));
  xa.L = L; xa.tpr = tpr; xa.nl = nl; xa.Sp_tot = p.Sp_tot; xa.rsp = c->w->p_sp.as<float>();
  int nkb = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {
    xa.map[j] = p.map[j]; xa.rows[j] = p.U[j] * p.Sp[j]; xa.S[j] = p.S[j]; xa.Sp[j] = p.Sp[j]; xa.off[j] = p.off[j];
    xa.cbt[j] = cbt_now[j]; xa.att[j] = p.att[j]; xa.att_slot[j] = p.att_slot[j];
    memcpy(xa.inst[j], p.rt_inst[j], RT_ARG_ROWS);
    xa.blk0[j] = nkb; nkb += p.Sp[j] / 16;
  }
  xa.blk0[CFD_NMEM] = nkb;
  xa.att_b0 = p.att_nb > 0 ? p.att_b0 : 0;
  xa.att_nb = p.att_nb > 0 ? p.att_nb : p.Be;
  xa.att_step = p.att_nb > 0 ? c->w->d_step.as<int>() : nullptr;
  for (int l = 0; l < nl; ++l) {
    const LayerW& w = c->lw[l];
    char* qk = sv ? sv->qk[l] : c->w->qk_sp.as<char>();
    char* vt = sv ? sv->vt[l] : c->w->rt_vt.as<char>();
    // ---- a. self attention: x += Wo softmax(q k^T) v                         (cross_attention.py:568-572)
    {
      RtGemmArgs a = base;   // norm1 + q | k | v^T projections
      a.x = X(l, 0);
      a.g = w.ln1g; a.b = w.ln1b; a.w = w.wqk_sp.as<char>(); a.w2 = w.wv_sp.as<char>(); a.nfb_qk = 2 * CFD_D / 16;
      a.bias = w.bqk.as<float>(); a.o_sp = qk; a.ld_o = 2 * CFD_D * 4; a.vt = vt;
      RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_LN, RT_EPI_QKV, 512, CFD_D / 32, 3, 3 * CFD_D, a);
    }
    {
      RtSelfArgs a{qk, vt, c->w->o_sp.as<char>(), L, tpr};
      Bracket br(c, CFD_PROF_GEMM_ATTN, st);
      hipLaunchKernelGGL(rt_selfattn_kernel<>, dim3(CFD_NHEAD, ntile), dim3(256), 40960, st, a);
      HIPCHK(hipGetLastError());
    }
    {
      RtGemmArgs a = base;   // out-projection + residual
      a.xr = X(l, 0); a.xo = X(l, 1);
      a.a_sp = c->w->o_sp.as<char>(); a.w = w.wo_sp.as<char>(); a.bias = w.bo2.as<float>();
      if (nfb2) RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 2, CFD_D, a);
      else RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 1, CFD_D, a);
    }
    if (c->stop_stage == 2 + 4 * l) return CFD_OK;
    // ---- b. time block 1                                                        (:575, :426-439)
    CHK(time_block(w.wtb1_sp, w.tb1g, w.tb1b, w.btb1, 2 * l, X(l, 1), X(l, 2)));
    if (c->stop_stage == 3 + 4 * l) {   // (test hook: the tap is read from x)
      if (!sv) HIPCHK(hipMemcpyAsync(xw, hw, (size_t)p.M * CFD_D * 4, hipMemcpyDeviceToDevice, st));
      return CFD_OK;
    }
    // ---- c-e. five cross attentions + fuser, folded                             (:578-652)
    {
      RtXArgs a = xa;
      a.x = X(l, 2); a.xo = X(l, 3);
      a.sc = sv ? sv->sc[l] : c->w->sc.as<float>();
      a.cst = sv ? sv->cst[l] : c->w->ssc.as<float>();   // (the tile-kernel path's self-attention score buffer: >= M x 32 float4, unused here)
      a.ln_g = w.ln2g; a.ln_b = w.ln2b; a.bias = w.cross_bias.as<float>(); a.layer = l;
      for (int j = 0; j < CFD_NMEM; ++j) {
        const size_t rows = (size_t)p.U[j] * p.Sp[j];
        a.K[j] = c->w->kall_sp[j].as<char>() + (size_t)l * rows * CFD_D * 4;
        a.VT[j] = c->w->vt_all[j].as<char>() + (size_t)l * rows * CFD_D * 4;
        a.kb[j] = kb_now[j] + (size_t)l * CFD_D;
        a.vb[j] = vb_now[j] + (size_t)l * CFD_D;
      }
      {
        Bracket br(c, CFD_PROF_XATTN, st);
        hipLaunchKernelGGL(rt_xscore_kernel<>, dim3(p.Sp_tot / 32, ntile), dim3(512), RT_XS_LDS, st, a);
        HIPCHK(hipGetLastError());
      }
      {
        Bracket br(c, CFD_PROF_XATTN, st);
        if (p.Sp_tot <= 512) hipLaunchKernelGGL(rt_xpv_kernel<512>, dim3(CFD_D / 16, ntile), dim3(512), lds_xpv, st, a);
        else hipLaunchKernelGGL(rt_xpv_kernel<RT_MAX_KEYS>, dim3(CFD_D / 16, ntile), dim3(512), lds_xpv, st, a);
        HIPCHK(hipGetLastError());
      }
    }
    if (c->stop_stage == 4 + 4 * l) return CFD_OK;
    if (sv && l == nl - 1) return CFD_OK;
    // ---- f. time block 2                                                        (:655)
    CHK(time_block(w.wtb2_sp, w.tb2g, w.tb2b, w.btb2, 2 * l + 1, X(l, 3), X(l, 4)));
    // ---- g. FFN                                                                 (:659-661)
    {
      RtGemmArgs a = base;   // norm3 + linear1 + GELU
      a.x = X(l, 4);
      a.g = w.ln3g; a.b = w.ln3b; a.w = w.w1_sp.as<char>(); a.bias = w.b1; a.o_sp = c->w->u_sp.as<char>(); a.ld_o = CFD_FF * 4; a.gelu = 1;
      a.pre = sv ? sv->pre[l] : nullptr;
      RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_LN, RT_EPI_SPLIT, 512, CFD_D / 32, 2, CFD_FF, a);
    }
    {
      RtGemmArgs a = base;   // linear2 + residual
      a.xr = X(l, 4); a.xo = X(l + 1, 0);
      a.a_sp = c->w->u_sp.as<char>(); a.w = w.w2_sp.as<char>(); a.bias = w.b2;
      RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_RESID, 512, CFD_FF / 32, 1, CFD_D, a);
    }
    if (c->stop_stage == 5 + 4 * l) return CFD_OK;
  }
  // 7. final norm + latent projection                                (cross_attention.py:238-239, denoiser.py:382)
  {
    RtGemmArgs a = base;
    a.x = X(nl, 0);
    a.g = rawp(c, "decoder.norm.weight"); a.b = rawp(c, "decoder.norm.bias"); a.w = c->wp_sp.as<char>();
    a.bias = rawp(c, "latent_proj.bias"); a.o_f32 = c->w->eps.as<float>(); a.ldo_f = CFD_LAT;
    RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_LN, RT_EPI_F32, 512, CFD_D / 32, 1, CFD_LAT, a);
  }
#undef RT_LAUNCH
  return CFD_OK;
}

int enqueue_rows(Ctx* c, hipStream_t st, int row0, int Be) {
  const Problem& p = c->w->pb;
  const int nl = c->nl, L = p.L;
  const long long M = (long long)Be * L;
  const int* dstep = p.tmode ? c->w->d_step.as<int>() + 1 : c->w->d_step.as<int>();
  const long long ROWB = CFD_D * 4;  // bytes per SP row of 512
  const dim3 blk(256);
  const char* sample_sp = c->w->sample_sp.as<char>() + (size_t)row0 * L * CFD_LAT * 4;
  float* eps_out = c->w->eps.as<float>() + (size_t)row0 * L * CFD_LAT;
  const int* mapj[CFD_NMEM];
  float* attj[CFD_NMEM];
  bool want_att = false;
  for (int j = 0; j < CFD_NMEM; ++j) {
    mapj[j] = p.map[j] + row0;
    attj[j] = p.att[j] ? p.att[j] + (size_t)row0 * nl * L * p.S[j] : nullptr;
    want_att = want_att || p.att[j];
  }
  if (p.att_fused) want_att = false;   // (the ring of a sampling run on the tile kernels: the fused kernel keeps the maps itself)
  // one fused kernel per layer for the cross-attention block, unless att_mats are wanted (or the naive debug GEMMs)
  const bool fused_x = c->fused_xattn && p.xa_nwg > 0 && !want_att && !g_cfd_naive_gemm && row0 == 0 && Be == p.Be;

  // rows that run the replica-independent head of the network (see Problem::share_B)
  const bool share = p.share_B > 0 && row0 == 0 && Be == p.Be && Be % p.share_B == 0 && Be > p.share_B && !c->stop_stage;
  const int Bs = share ? p.share_B : Be;
  const long long Ms = (long long)Bs * L;
  // 1. latent embedding + body/hand embedding + query PE          (denoiser.py:187,316-326)
  {
    GemmArgs a = gemm_args();
    a.X[0] = c->we_sp.as<char>(); a.ldx[0] = CFD_LAT * 4; a.I[0] = CFD_D; a.Iclamp[0] = CFD_D; a.kt[0] = CFD_LAT / 32;
    a.Y = sample_sp; a.ldy = CFD_LAT * 4; a.J = (int)Ms; a.Jclamp = (int)Ms;
    EpiEmbed e{c->w->x.as<float>(), rawp(c, "latent_embd.bias"), rawp(c, "bh_embedding.weight"), rawp(c, "query_pos.pe"), L};
    CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
  }
  if (c->stop_stage == 1) return CFD_OK;
  // This step's rows of the per-step tables (as on the row-tile path, enqueue_rows_rt): with one timestep for all rows, a sampling run
  // refreshes fixed buffers with ONE launch at the start of the iteration and cfd_forward points at its single table row, so that no
  // launch has the step index as a dependent scalar load in front of its operand loads (a load that misses in every XCD's L2 after the
  // previous iteration's last workgroup has written it: ~1 us on 18 AdaLN launches and 9 cross-attention prologues per step).
  const float* ss_now = c->w->ss_tab.as<float>();
  const float *kb_now[CFD_NMEM], *vb_now[CFD_NMEM];
  for (int j = 0; j < CFD_NMEM; ++j) { kb_now[j] = c->w->kbtab[j].as<float>(); vb_now[j] = c->w->vbtab[j].as<float>(); }
  const bool rows_now = p.tmode == 0 && c->step_rows;
  if (rows_now && p.T > 1) {
    RtStepRowsArgs ra;
    memset(&ra, 0, sizeof(ra));
    size_t off4 = 0;
    int nt = 0, nwg = 0;
    auto add = [&](const float*& now, int nfloat) {
      float* dst = c->w->rt_cur.as<float>() + off4 * 4;
      ra.src[nt] = now; ra.dst[nt] = dst; ra.n4[nt] = nfloat / 4; ra.first[nt] = nwg;
      nwg += (nfloat / 4 + 255) / 256; off4 += (size_t)(nfloat / 4 + 63) / 64 * 64; ++nt;
      now = dst;
    };
    size_t need4 = (size_t)(nl * 4 * CFD_D / 4 + 64);
    for (int j = 0; j < CFD_NMEM; ++j) need4 += (size_t)((nl * CFD_D + 32) / 4 + 64) + (size_t)(nl * CFD_D / 4 + 64);
    CHK(c->w->rt_cur.ensure(need4 * 16));
    add(ss_now, nl * 4 * CFD_D);
    for (int j = 0; j < CFD_NMEM; ++j)
      if ((p.static_mask >> j) & 1) { add(kb_now[j], nl * CFD_D + 32); add(vb_now[j], nl * CFD_D); }
    ra.ntab = nt; ra.first[nt] = nwg; ra.d_step = dstep;
    LAUNCH(CFD_PROF_OTHER, rt_step_rows_kernel<>, dim3(nwg), dim3(256), st, ra);
  }
  auto ln = [&](const float* g, const float* b, int adaln, int tbidx, char* out, long long rows) -> int {
    LnArgs a{c->w->x.as<float>(), out, rows, g, b, adaln, (rows_now ? ss_now : c->w->ss_tab.as<float>()) + (size_t)tbidx * 2 * CFD_D,
             (long long)nl * 2 * 2 * CFD_D, rows_now ? nullptr : dstep, p.tmode, L, row0};
    LAUNCH(CFD_PROF_ROWS, ln_rows_kernel<>, dim3((unsigned)((rows + 3) / 4)), blk, st, a);
    return CFD_OK;
  };
  auto token_gemm_resid = [&](const DBuf& w, int K, const char* y, const float* bias, long long rows) -> int {
    GemmArgs a = gemm_args();
    a.X[0] = w.as<char>(); a.ldx[0] = (long long)K * 4; a.I[0] = CFD_D; a.Iclamp[0] = CFD_D; a.kt[0] = K / 32;
    a.Y = y; a.ldy = (long long)K * 4; a.J = (int)rows; a.Jclamp = (int)rows;
    EpiResid e{c->w->x.as<float>(), 0, bias};
    return run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st);
  };
  // residual product + the LayerNorm that follows it
  auto token_gemm_resid_ln = [&](const DBuf& w, int K, const char* y, const float* bias, long long rows, const float* g, const float* b,
                                 int adaln, int tbidx) -> int {
    CHK(token_gemm_resid(w, K, y, bias, rows));
    return ln(g, b, adaln, tbidx, c->w->h_sp.as<char>(), rows);
  };
  // The algebraic LayerNorm fold (gemm_sp.hpp EpiLn / EpiResidStat) for mid-size problems at the product shape: norm3, the
  // norm1 of layers 1.. and the decoder's final norm are not launched; the residual product in front of each stores the split pairs of the RAW rows and
  // per-row slot statistics, and the consumer (q | k | v^T in one launch, FFN1, latent_proj) runs on W diag(gamma) and rescales its accumulators.
  // Range: the launches launch_gemm gives the 64 x 64 / 128 x 64 classes anyway (launch_gemm_midsize); above it a ln_rows launch costs less than
  // the producer's wider epilogue (measured at the headline shape: +25.8 us against 14).
  const bool ln_fold = c->ln_fold != 0 && L == 16 && c->qkv_fused && !g_cfd_naive_gemm && !c->stop_stage && row0 == 0 && Be == p.Be &&
                       (c->ln_fold > 0 || (M >= 512 && M <= 3840));   // (below: the row-tile path or a handful of workgroups; above: launch_gemm's 128 x 128 class)
  if (ln_fold) CHK(c->w->ln_stat.ensure((size_t)M * LN_SLOTS * 2 * 4));
  auto token_gemm_resid_stat = [&](const DBuf& w, int K, const char* y, const float* bias, long long rows, char* xs) -> int {
    GemmArgs a = gemm_args();
    a.X[0] = w.as<char>(); a.ldx[0] = (long long)K * 4; a.I[0] = CFD_D; a.Iclamp[0] = CFD_D; a.kt[0] = K / 32;
    a.Y = y; a.ldy = (long long)K * 4; a.J = (int)rows; a.Jclamp = (int)rows;
    EpiResidStat e{c->w->x.as<float>(), 0, bias, xs, c->w->ln_stat.as<float>()};
    return run_gemm_midsize<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, st);
  };
  bool h_raw = false;     // h_sp holds the split pairs of the raw rows + ln_stat their statistics (ln_fold), not norm1(x)
  static unsigned long long attr = 0;   // per device (one bit per ordinal): a process may hold handles on several GPUs
  if (!((attr >> (c->cfg.device & 63)) & 1ull)) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn_fused_kernel<>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
#if XA_ALL_OPF
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 11>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
#endif
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 15 | XA_DBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    attr |= 1ull << (c->cfg.device & 63);
  }

  bool h_ready = false;   // h_sp already holds norm1(x) of the layer that starts
  for (int l = 0; l < nl; ++l) {
    const LayerW& w = c->lw[l];
    // rows of sub-layers a and b: layer 0 runs them once per utterance when the batch is G replicas of it
    const int Ba = (l == 0) ? Bs : Be;
    const long long Ma = (long long)Ba * L;
    // ---- a. self attention: x += Wo softmax(q k^T) v                         (cross_attention.py:568-572)
    // (norm1 of layers 1.. is made by the previous layer's last residual product when that ran row-complete: `h_ready`)
    if (!h_ready) CHK(ln(w.ln1g, w.ln1b, 0, 0, c->w->h_sp.as<char>(), Ma));
    h_ready = false;
    bool qkv_one_launch = false;
    const int Lv = (L + 31) / 32 * 32;   // (whole 32-key blocks: at L = 16 a 64-key pitch made the v^T product twice the work of the q | k one)
    {
      // q (pre-scaled) and k, token-major ...
      GemmArgs a = gemm_args();
      a.X[0] = w.wqk_sp.as<char>(); a.ldx[0] = ROWB; a.I[0] = 2 * CFD_D; a.Iclamp[0] = 2 * CFD_D; a.kt[0] = CFD_D / 32;
      a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.J = (int)Ma; a.Jclamp = (int)Ma;
      EpiSplit e{c->w->qk_sp.as<char>(), 2 * ROWB, 0, 0, w.bqk.as<float>(), 0, 0};
      // ... and v^T per batch row: vts[b][f][l] (keys in P-fragment order for the fused kernel)
      GemmArgs av = gemm_args();
      av.X[0] = c->w->h_sp.as<char>(); av.ldx[0] = ROWB; av.xbs[0] = (long long)L * ROWB; av.I[0] = Lv; av.Iclamp[0] = L; av.kt[0] = CFD_D / 32;
      av.Y = w.wv_sp.as<char>(); av.ldy = ROWB; av.J = CFD_D; av.Jclamp = CFD_D;
      EpiSplit ev{c->w->vts_sp.as<char>(), (long long)Lv * 4, (long long)CFD_D * Lv * 4, 0, nullptr, 0, 1};
      if (L == 16 && c->qkv_fused && !g_cfd_naive_gemm) {
        // batch rows of exactly 16 tokens: both in ONE grouped launch, the value projection stored transposed by the epilogue (EpiQkvT)
        GemmArgs ag = a;
        ag.nslot = 2;
        ag.X[1] = w.wv_sp.as<char>(); ag.ldx[1] = ROWB; ag.I[1] = CFD_D; ag.Iclamp[1] = CFD_D; ag.kt[1] = CFD_D / 32;
        EpiQkvT eg{c->w->qk_sp.as<char>(), 2 * ROWB, w.bqk.as<float>(), c->w->vts_sp.as<char>(), c->qkv_fused == 1 ? 1 : 0};
        if (h_raw) {   // (layers 1..: the previous layer's second FFN product left raw rows + statistics)
          EpiLn<EpiQkvT> el;
          static_cast<EpiQkvT&>(el) = eg;
          const float* cd = w.ln_cd.as<float>();
          el.ln_stat = c->w->ln_stat.as<float>(); el.ln_c[0] = cd; el.ln_d[0] = cd + 1024; el.ln_c[1] = cd + 2048; el.ln_d[1] = cd + 2560; el.ln_eps = 1e-5f;
          ag.X[0] = w.wqk_f.as<char>(); ag.X[1] = w.wv_f.as<char>();
          CHK((run_gemm_midsize<MODE_GROUPED>(c, CFD_PROF_GEMM_TOKEN, ag, el, st)));
        } else
        CHK((run_gemm<MODE_GROUPED>(c, CFD_PROF_GEMM_TOKEN, ag, eg, 1, 1, st)));
        qkv_one_launch = true;
      } else {
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, av, ev, Ba, 1, st)));
      }
    }
    if (qkv_one_launch && c->qkv_fused == 1) {
      // one query tile and one key tile per (row, head): the row-tile path's attention core (4 waves that all compute; V^T in natural key
      // order, which EpiQkvT wrote) instead of the flash kernel's 8-wave workgroup with one busy wave
      RtSelfArgs a{c->w->qk_sp.as<char>(), c->w->vts_sp.as<char>(), c->w->o_sp.as<char>(), L, 1};
      Bracket br(c, CFD_PROF_GEMM_ATTN, st);
      hipLaunchKernelGGL(rt_selfattn_kernel<>, dim3(CFD_NHEAD, Ba), dim3(256), 40 * 1024, st, a);
      HIPCHK(hipGetLastError());
    } else {
      SelfAttnArgs a{c->w->qk_sp.as<char>(), c->w->vts_sp.as<char>(), c->w->o_sp.as<char>(), L, Lv};
      Bracket br(c, CFD_PROF_GEMM_ATTN, st);
      hipLaunchKernelGGL(self_attn_fused_kernel<>, dim3((L + SELF_ATTN_WAVES * 16 - 1) / (SELF_ATTN_WAVES * 16), CFD_NHEAD, Ba), dim3(SELF_ATTN_WAVES * 64), 65536, st, a);
      HIPCHK(hipGetLastError());
    }
    // out-projection + residual, then time block 1's AdaLN + SiLU                 (:572, :575, :426-439)
    CHK(token_gemm_resid_ln(w.wo_sp, CFD_D, c->w->o_sp.as<char>(), w.bo2.as<float>(), Ma, w.tb1g, w.tb1b, 1, 2 * l));
    if (c->stop_stage == 2 + 4 * l) return CFD_OK;
    // ---- b. time block 1
    CHK(token_gemm_resid(w.wtb1_sp, CFD_D, c->w->h_sp.as<char>(), w.btb1, Ma));
    if (c->stop_stage == 3 + 4 * l) return CFD_OK;
    if (Ma != M) {   // every guidance chunk starts its first cross-attention from the same state
      const long long n4 = Ma * (CFD_D / 4);
      LAUNCH(CFD_PROF_ROWS, replicate_rows_kernel<>, dim3((unsigned)((n4 + 255) / 256)), blk, st, reinterpret_cast<float4*>(c->w->x.as<float>()), n4,
             (int)(M / Ma));
    }
    // ---- c-e. five cross attentions + fuser, folded                             (:578-652)
    if (fused_x) {   // LayerNorm2 is part of the kernel's prologue
      XAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.x = c->w->x.as<float>(); a.ln_g = w.ln2g; a.ln_b = w.ln2b; a.bias = w.cross_bias.as<float>(); a.L = L;
      for (int j = 0; j < CFD_NMEM; ++j) {
        const size_t rows = (size_t)p.U[j] * p.Sp[j];
        // (single-fp16 tiles of a long memory, Problem::xa_opf / xa_f16_mask: 32 KB per 32 keys = 1 KB per key, tile-major per (layer, instance))
        const bool f16 = (p.xa_f16_mask >> j) & 1;
        a.K[j] = (f16 && (p.xa_opf & XA_K16)) ? c->w->k16[j].as<char>() + (size_t)l * rows * 1024 : c->w->kall_sp[j].as<char>() + (size_t)l * rows * ROWB;
        a.cb[j] = c->w->cb[j].as<float>() + (size_t)l * rows;
        a.VT[j] = (f16 && (p.xa_opf & XA_V16)) ? c->w->v16[j].as<char>() + (size_t)l * rows * 1024 : c->w->vt_all[j].as<char>() + (size_t)l * rows * ROWB;
        a.Sp[j] = p.Sp[j];
        const bool stat = (p.static_mask >> j) & 1;
        a.rs_off[j] = (unsigned)((size_t)(nl - l) * rows * 4);
        a.kb[j] = stat ? kb_now[j] + (size_t)l * CFD_D : c->w->zeros512.as<float>();
        a.kb_stride[j] = stat ? nl * CFD_D + 32 : 0;
        a.vb[j] = stat ? vb_now[j] + (size_t)l * CFD_D : c->w->zeros512.as<float>();
        a.vb_stride[j] = stat ? nl * CFD_D : 0;
      }
      a.d_step = rows_now ? nullptr : dstep;
      a.one_j = -1;
      if (p.xa_one >= 0) {   // (the work lists hold no segments for it: build_xattn_worklist)
        const int j = p.xa_one;
        const size_t rows = (size_t)p.U[j] * p.Sp[j];
        a.one_j = j; a.one_sp = p.Sp[j];
        a.one_va = c->w->xa_one_va.as<float>() + (size_t)l * p.U[j] * CFD_D;
        a.one_rs = c->w->cb[j].as<float>() + (size_t)nl * rows;
      }
      a.wgs = c->w->xa_wgs.as<XaWg>(); a.segs = c->w->xa_segs.as<XaSeg>();
#if XA_STAMP
      CHK(c->w->xa_stamps.ensure((size_t)p.xa_nwg * XA_WAVES * XA_NSTAMP * 8));
      a.stamps = c->w->xa_stamps.as<long long>();
#endif
      Bracket br(c, CFD_PROF_XATTN, st);
      if (p.att_fused) a.att = c->w->xa_att_desc.as<XaAtt>() + l;
      const int opf = p.att_fused ? 0 : p.xa_opf;
      const bool xa_db = XA_ALL_OPF ? c->xa_db != 0 : true;   // (developer builds: CFD_XA_DB=0 puts OPF 15 back on the three-barrier step)
      (void)xa_db;
      auto launch_xa = [&](int nwg, const XAttnArgs& xa) {
        if (p.att_fused) hipLaunchKernelGGL((xattn_fused_kernel<true, 0>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
#if XA_ALL_OPF
        else if (opf == 1) hipLaunchKernelGGL((xattn_fused_kernel<false, 1>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 2) hipLaunchKernelGGL((xattn_fused_kernel<false, 2>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 3) hipLaunchKernelGGL((xattn_fused_kernel<false, 3>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 7) hipLaunchKernelGGL((xattn_fused_kernel<false, 7>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 11) hipLaunchKernelGGL((xattn_fused_kernel<false, 11>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 15 && !xa_db) hipLaunchKernelGGL((xattn_fused_kernel<false, 15>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
#endif
        else if (opf == 15) hipLaunchKernelGGL((xattn_fused_kernel<false, 15 | XA_DBUF>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else hipLaunchKernelGGL((xattn_fused_kernel<false, 0>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
      };
      if (l == 0 && share && p.xa0_nwg_a > 0) {   // layer-0 de-duplication (build_xattn_layer0_lists): the longest memory once per distinct (utterance, instance) ...
        XAttnArgs a0 = a;
        a0.wgs = c->w->xa0_wgs_a.as<XaWg>(); a0.segs = c->w->xa0_segs_a.as<XaSeg>(); a0.dd_out = c->w->xa_dedup.as<float>(); a0.stamps = nullptr;
        a0.one_j = -1;   // (the one-key memory belongs to the second launch)
        launch_xa(p.xa0_nwg_a, a0);
        HIPCHK(hipGetLastError());
        // ... then the other memories for every row, which also adds the stored results
        a0.wgs = c->w->xa0_wgs_b.as<XaWg>(); a0.segs = c->w->xa0_segs_b.as<XaSeg>(); a0.dd_out = nullptr; a0.dd_in = c->w->xa_dedup.as<float>();
        a0.one_j = a.one_j;
        launch_xa(p.xa0_nwg_b, a0);
        HIPCHK(hipGetLastError());
        if (c->prof) c->prof_n[CFD_PROF_XATTN] += 1;   // (two launches under one bracket)
      } else {
        launch_xa(p.xa_nwg, a);
        HIPCHK(hipGetLastError());
      }
    } else {
    CHK(ln(w.ln2g, w.ln2b, 0, 0, c->w->h_sp.as<char>(), M));
    // Three-launch path (att_mats wanted): scores against the folded keys of every memory.  Long memories and short
    // (<= 64 keys) memories go to different tile shapes; rows in a shared-memory run of the largest memory use one
    // un-batched product per run.
    const bool runs = p.nruns > 0 && row0 == 0 && Be == p.Be;
    auto scores_grouped = [&](bool small, int skip_j, const int* brow, int nb) -> int {
      GemmArgs a = gemm_args();
      EpiF32 e;
      memset(&e, 0, sizeof(e));
      int n = 0;
      for (int j = 0; j < CFD_NMEM; ++j) {
        if ((p.Sp[j] <= 64) != small || j == skip_j) continue;
        a.X[n] = c->w->kall_sp[j].as<char>() + (size_t)l * p.U[j] * p.Sp[j] * ROWB; a.ldx[n] = ROWB;
        a.xbs[n] = (long long)p.Sp[j] * ROWB; a.xmap[n] = mapj[j];
        a.I[n] = p.Sp[j]; a.Iclamp[n] = p.Sp[j]; a.kt[n] = CFD_D / 32;
        e.goff[n] = p.off[j]; e.gbias[n] = c->w->cb[j].as<float>() + (size_t)l * p.U[j] * p.Sp[j]; e.gmap[n] = mapj[j]; e.gstride[n] = p.Sp[j];
        ++n;
      }
      if (!n || nb <= 0) return CFD_OK;
      a.nslot = n; a.brow = brow;
      a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.ybs = (long long)L * ROWB; a.J = L; a.Jclamp = L;
      e.out = c->w->sc.as<float>(); e.ldo = p.Sp_tot; e.obs = (long long)L * p.Sp_tot;
      return run_gemm<MODE_GROUPED>(c, CFD_PROF_GEMM_ATTN, a, e, nb, 1, st);
    };
    CHK(scores_grouped(true, -1, nullptr, Be));   // short memories
    if (!runs) {
      CHK(scores_grouped(false, -1, nullptr, Be));
    } else {
      CHK(scores_grouped(false, -1, c->w->short_rows.as<int>(), p.nshort));
      CHK(scores_grouped(false, p.jbig, c->w->long_rows.as<int>(), p.nlong));
      const int j = p.jbig;
      for (int r = 0; r < p.nruns; ++r) {
        GemmArgs a = gemm_args();
        a.X[0] = c->w->kall_sp[j].as<char>() + ((size_t)l * p.U[j] + p.run_u[r]) * p.Sp[j] * ROWB; a.ldx[0] = ROWB;
        a.I[0] = p.Sp[j]; a.Iclamp[0] = p.Sp[j]; a.kt[0] = CFD_D / 32;
        a.Y = c->w->h_sp.as<char>() + (size_t)p.run_row0[r] * L * ROWB; a.ldy = ROWB; a.J = p.run_len[r] * L; a.Jclamp = a.J;
        EpiF32 e;
        memset(&e, 0, sizeof(e));
        e.out = c->w->sc.as<float>() + (size_t)p.run_row0[r] * L * p.Sp_tot; e.ldo = p.Sp_tot; e.goff[0] = p.off[j];
        e.gbias[0] = c->w->cb[j].as<float>() + ((size_t)l * p.U[j] + p.run_u[r]) * p.Sp[j]; e.gstride[0] = 0;
        CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_ATTN, a, e, 1, 1, st)));
      }
    }
    {
      SoftmaxArgs a;
      memset(&a, 0, sizeof(a));
      a.sc = c->w->sc.as<float>(); a.P = c->w->p_sp.as<char>(); a.ld = p.Sp_tot; a.rows = M; a.rows_per_b = L; a.nseg = CFD_NMEM;
      for (int j = 0; j < CFD_NMEM; ++j) {
        a.off[j] = p.off[j]; a.S[j] = p.S[j]; a.Sp[j] = p.Sp[j]; a.mask[j] = p.mask[j]; a.has_mask[j] = p.has_mask[j]; a.map[j] = mapj[j]; a.att[j] = attj[j];
      }
      a.layer = l; a.nl = nl;
      LAUNCH(CFD_PROF_ROWS, softmax_rows_kernel<>, dim3((unsigned)((M + 3) / 4)), blk, st, a);
    }
    // x += sum_j P_j . VV_j(n_j) + folded bias
    auto pv_segk = [&](int skip_j, const int* brow, int nb) -> int {
      if (nb <= 0) return CFD_OK;
      GemmArgs a = gemm_args();
      int n = 0;
      for (int j = 0; j < CFD_NMEM; ++j) {
        if (j == skip_j) continue;
        a.X[n] = c->w->vt_all[j].as<char>() + (size_t)l * p.U[j] * CFD_D * p.Sp[j] * 4; a.ldx[n] = (long long)p.Sp[j] * 4;
        a.xbs[n] = (long long)CFD_D * p.Sp[j] * 4; a.xmap[n] = mapj[j];
        a.kt[n] = p.Sp[j] / 32; a.yk0[n] = p.off[j] / 32;
        a.I[n] = CFD_D; a.Iclamp[n] = CFD_D;
        ++n;
      }
      a.nslot = n; a.brow = brow;
      a.Y = c->w->p_sp.as<char>(); a.ldy = (long long)p.Sp_tot * 4; a.ybs = (long long)L * p.Sp_tot * 4; a.J = L; a.Jclamp = L;
      EpiResid e{c->w->x.as<float>(), (long long)L * CFD_D, w.cross_bias.as<float>()};
      return run_gemm<MODE_SEGK>(c, CFD_PROF_GEMM_ATTN, a, e, nb, 1, st);
    };
    if (!runs) {
      CHK(pv_segk(-1, nullptr, Be));
    } else {
      CHK(pv_segk(-1, c->w->short_rows.as<int>(), p.nshort));   // short rows: all segments
      CHK(pv_segk(p.jbig, c->w->long_rows.as<int>(), p.nlong));   // long rows: short segments first ...
      const int j = p.jbig;
      for (int r = 0; r < p.nruns; ++r) {   // ... then the shared audio memory, run by run (disjoint rows)
        GemmArgs a = gemm_args();
        a.X[0] = c->w->vt_all[j].as<char>() + ((size_t)l * p.U[j] + p.run_u[r]) * CFD_D * p.Sp[j] * 4; a.ldx[0] = (long long)p.Sp[j] * 4;
        a.I[0] = CFD_D; a.Iclamp[0] = CFD_D; a.kt[0] = p.Sp[j] / 32;
        a.Y = c->w->p_sp.as<char>() + (size_t)p.run_row0[r] * L * p.Sp_tot * 4 + (size_t)(p.off[j] / 32) * 128;
        a.ldy = (long long)p.Sp_tot * 4; a.J = p.run_len[r] * L; a.Jclamp = a.J;
        EpiResid e{c->w->x.as<float>() + (size_t)p.run_row0[r] * L * CFD_D, 0, nullptr};
        CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_ATTN, a, e, 1, 1, st)));
      }
    }
    }
    if (c->stop_stage == 4 + 4 * l) return CFD_OK;
    // ---- f. time block 2                                                        (:655)
    CHK(ln(w.tb2g, w.tb2b, 1, 2 * l + 1, c->w->h_sp.as<char>(), M));
    // time block 2's projection + residual, then norm3                           (:655, :659)
    if (ln_fold) CHK(token_gemm_resid_stat(w.wtb2_sp, CFD_D, c->w->h_sp.as<char>(), w.btb2, M, c->w->o_sp.as<char>()));   // (o_sp: free since the out-projection)
    else CHK(token_gemm_resid_ln(w.wtb2_sp, CFD_D, c->w->h_sp.as<char>(), w.btb2, M, w.ln3g, w.ln3b, 0, 0));
    // ---- g. FFN                                                                 (:659-661)
    {
      GemmArgs a = gemm_args();
      a.X[0] = w.w1_sp.as<char>(); a.ldx[0] = ROWB; a.I[0] = CFD_FF; a.Iclamp[0] = CFD_FF; a.kt[0] = CFD_D / 32;
      a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.J = (int)M; a.Jclamp = (int)M;
      EpiSplit e{c->w->u_sp.as<char>(), (long long)CFD_FF * 4, 0, 0, w.b1, 1, 0};
      if (ln_fold) {
        EpiLn<EpiSplit> el;
        static_cast<EpiSplit&>(el) = e;
        const float* cd = w.ln_cd.as<float>();
        el.ln_stat = c->w->ln_stat.as<float>(); el.ln_c[0] = el.ln_c[1] = cd + 3072; el.ln_d[0] = el.ln_d[1] = cd + 4096; el.ln_eps = 1e-5f;
        a.X[0] = w.w1_f.as<char>(); a.Y = c->w->o_sp.as<char>();
        CHK((run_gemm_midsize<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, el, st)));
      } else
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
    }
    // second FFN product + residual, then the next layer's norm1 (or the decoder's final norm)   (:661, :568; :238-239)
    {
      const float* ng = l + 1 < nl ? c->lw[l + 1].ln1g : rawp(c, "decoder.norm.weight");
      const float* nb = l + 1 < nl ? c->lw[l + 1].ln1b : rawp(c, "decoder.norm.bias");
      if (c->stop_stage == 5 + 4 * l) return token_gemm_resid(w.w2_sp, CFD_FF, c->w->u_sp.as<char>(), w.b2, M);
      if (ln_fold) { CHK(token_gemm_resid_stat(w.w2_sp, CFD_FF, c->w->u_sp.as<char>(), w.b2, M, c->w->h_sp.as<char>())); h_raw = true; }
      else CHK(token_gemm_resid_ln(w.w2_sp, CFD_FF, c->w->u_sp.as<char>(), w.b2, M, ng, nb, 0, 0));
      h_ready = true;
    }
  }
  // 7. final norm (made above) + latent projection                                (cross_attention.py:238-239, denoiser.py:382)
  {
    GemmArgs a = gemm_args();
    a.X[0] = c->wp_sp.as<char>(); a.ldx[0] = ROWB; a.I[0] = CFD_LAT; a.Iclamp[0] = CFD_LAT; a.kt[0] = CFD_D / 32;
    a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.J = (int)M; a.Jclamp = (int)M;
    EpiF32 e;
    memset(&e, 0, sizeof(e));
    e.out = eps_out; e.ldo = CFD_LAT; e.bias = rawp(c, "latent_proj.bias");
    if (h_raw) {
      EpiLn<EpiF32> el;
      static_cast<EpiF32&>(el) = e;
      const float* cd = c->ln_cd_p.as<float>();
      el.ln_stat = c->w->ln_stat.as<float>(); el.ln_c[0] = el.ln_c[1] = cd; el.ln_d[0] = el.ln_d[1] = cd + CFD_LAT; el.ln_eps = 1e-5f;
      a.X[0] = c->wp_f.as<char>();
      CHK((run_gemm_midsize<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, el, st)));
    } else
    CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
  }
  if (p.att_fused && fused_x) {   // this step's maps: from what the nine cross-attention launches kept, into slot *d_step of the ring
    XaFixArgs f;
    memset(&f, 0, sizeof(f));
    f.att = c->w->xa_att_desc.as<XaAtt>(); f.nl = nl; f.L = L; f.one_j = p.xa_one; f.d_step = c->w->d_step.as<int>();
    for (int j = 0; j < CFD_NMEM; ++j) { f.S[j] = p.S[j]; f.ring[j] = p.att[j]; f.slot[j] = p.att_slot[j]; }
    LAUNCH(CFD_PROF_ROWS, att_fixup_kernel<>, dim3((unsigned)(p.att_nb * L), nl), dim3(256), st, f);
  }
  return CFD_OK;
}

int run_gemm_plain_f32(Ctx* c, int cls, const GemmArgs& a, const EpiF32& e, int nb, int nz, hipStream_t st) {
  return run_gemm<MODE_PLAIN>(c, cls, a, e, nb, nz, st);
}

// ---- cfd_forward ---------------------------------------------------------------------------------------
extern "C" int cfd_forward(cfd_handle c, const float* sample, int Be, int L, const int32_t* timesteps, int n_t,
                           const cfd_memory mem[CFD_NUM_MEM], float* out, float* const att[CFD_NUM_MEM], void* stream) {
  if (!c) return fail(CFD_E_ARG, "null argument");
  c->hint_now = c->hint_same_mem;   // the promise covers THIS call only, however it ends
  c->hint_same_mem = false;
  if (!sample || !timesteps || !mem || !out) return fail(CFD_E_ARG, "null argument");
  if (c->run_open) return fail(CFD_E_STATE, "a sampling run is open on this handle");
  HIPCHK(hipSetDevice(c->cfg.device));
  CHK(settle_deferred_census(c));
  if (n_t != 1 && n_t != Be) return fail(CFD_E_ARG, "n_t must be 1 or Be");
  hipStream_t st = (hipStream_t)stream;
  const int tmode = (n_t == 1) ? 0 : 1;
  CHK(setup_problem(c, Be, L, mem, att, tmode, n_t));
  CHK(sat_begin(c, st));
  // tmode 0 reads table row d_step[0] which must be 0 outside a sampling run
  HIPCHK(hipMemsetAsync(c->w->d_step.p, 0, 16, st));
  CHK(build_time_tables(c, timesteps, n_t, st));
  // (the sample is split in front of the once-per-call projections, so that ONE wait reads the census of both)
  CHK(enqueue_to_split(c, CFD_PROF_OTHER, st, sample, c->w->sample_sp.as<char>(), c->w->pb.M, CFD_LAT, (long long)CFD_LAT, (long long)CFD_LAT * 4, c->sat_in()));
  {
    bool want_att = false;
    for (int j = 0; j < CFD_NMEM; ++j) want_att = want_att || (att && att[j]);
    const bool reuse = c->hint_now && c->w->pb.prev_same;
    CHK(prepare_static_memside(c, st, 0, want_att && !c->w->pb.att_fused, reuse));
    if (c->w->pb.static_mask) {   // once-per-call projections of the caller's memories: the census is read before they are used
      HIPCHK(hipStreamSynchronize(st));
      CHK(check_saturation(c, "cfd_forward (sample, memories / their projections)"));
    }
  }
  c->memside_in_forward = false;
  CHK(enqueue_denoise(c, st));
  HIPCHK(hipMemcpyAsync(out, c->w->eps.p, (size_t)c->w->pb.M * CFD_LAT * 4, hipMemcpyDeviceToDevice, st));
  if (c->memside_in_forward || !c->w->pb.static_mask) {
    // Paths whose memory-side projections run INSIDE the forward (per-row timesteps, att_mats on the tile kernels, CFD_HOIST_MEMSIDE=0,
    // the three-launch cross-attention): their census -- and the sample's, which no earlier wait has read on these paths -- is read here,
    // so that a clamped projection fails THIS call instead of the next one (on these paths the call therefore returns with `out` complete).
    HIPCHK(hipStreamSynchronize(st));
    c->memside_in_forward = false;
    CHK(check_saturation(c, "cfd_forward (sample, memories / their per-call projections)"));
  }
  {   // what the next call may reuse (cfd_forward_same_memories): all five memories' timestep-independent projections are in the workspace
    Work* w = c->w;
    const Problem& p = w->pb;
    w->fwd_mem_valid = p.tmode == 0 && p.static_mask == (1 << CFD_NMEM) - 1;
    w->fwd_wver = (unsigned long long)c->wver;
    w->fwd_Be = p.Be;
    w->fwd_L = p.L;
    w->fwd_att = false;
    for (int j = 0; j < CFD_NMEM; ++j) w->fwd_att = w->fwd_att || (att && att[j]);
    for (int j = 0; j < CFD_NMEM; ++j) {
      w->fwd_U[j] = p.U[j]; w->fwd_S[j] = p.S[j]; w->fwd_mask[j] = mem[j].key_padding_mask != nullptr; w->fwd_map[j] = mem[j].row_map != nullptr;
    }
  }
  return CFD_OK;
}

extern "C" int cfd_forward_same_memories(cfd_handle c) {
  if (!c) return fail(CFD_E_ARG, "null handle");
  c->hint_same_mem = true;
  return CFD_OK;
}

extern "C" int cfd_profile_forward(cfd_handle c, float ms[CFD_PROF_NCLASS], int launches[CFD_PROF_NCLASS]) {
  if (!c || !ms || !launches) return fail(CFD_E_ARG, "null argument");
  if (c->w->pb.Be == 0) return fail(CFD_E_STATE, "no problem configured (call cfd_forward or cfd_sample_begin first)");
  HIPCHK(hipSetDevice(c->cfg.device));
  if (c->run_open && c->run_pos >= c->run_iters)
    return fail(CFD_E_STATE, "sampling run is complete; profile before the last iteration");
  hipStream_t st = c->run_open ? c->run_stream : nullptr;
  HIPCHK(hipStreamSynchronize(st));
  for (int k = 0; k < CFD_PROF_NCLASS; ++k) { c->prof_ms[k] = 0.f; c->prof_n[k] = 0; }
  c->prof = true;
  int r = enqueue_denoise(c, st);
  c->prof = false;
  HIPCHK(hipStreamSynchronize(st));
  for (int k = 0; k < CFD_PROF_NCLASS; ++k) { ms[k] = c->prof_ms[k]; launches[k] = c->prof_n[k]; }
  return r;
}

// ---- developer hooks that share this unit's product instances (include/cfdenoise_dev.h) ---------------------
extern "C" int cfd_test_gemm(cfd_handle c, const float* X, const float* Y, float* out, int I, int J, int K, int tile_cfg,
                             void* stream) {
  if (!c || !X || !Y || !out || K % 32 || I % 4 || I < 4 || J < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  DBuf xs, ys;
  CHK(xs.ensure((size_t)I * K * 4));
  CHK(ys.ensure((size_t)J * K * 4));
  long long n = (long long)I * (K / 8);
  CHK(enqueue_to_split(c, CFD_PROF_OTHER, st, X, xs.as<char>(), (long long)I, K, (long long)K,
                     (long long)K * 4, nullptr));
  n = (long long)J * (K / 8);
  CHK(enqueue_to_split(c, CFD_PROF_OTHER, st, Y, ys.as<char>(), (long long)J, K, (long long)K,
                     (long long)K * 4, nullptr));
  GemmArgs a = gemm_args();
  a.X[0] = xs.as<char>(); a.ldx[0] = (long long)K * 4; a.I[0] = I; a.Iclamp[0] = I; a.kt[0] = K / 32;
  a.Y = ys.as<char>(); a.ldy = (long long)K * 4; a.J = J; a.Jclamp = J;
  EpiF32 e;
  memset(&e, 0, sizeof(e));
  e.out = out; e.ldo = I;
  hipError_t err = launch_gemm<MODE_PLAIN, EpiF32>(a, e, 1, 1, st, tile_cfg);
  if (err != hipSuccess) return fail(CFD_E_HIP, "gemm launch failed: %s", hipGetErrorString(err));
  HIPCHK(hipStreamSynchronize(st));
  xs.release();
  ys.release();
  return CFD_OK;
}

// Micro-benchmark hook: `iters` launches of the MFMA GEMM (EpiResid epilogue: x[j][i] += D + bias) on device-resident
// SP operands filled from a float32 pattern; returns the average milliseconds per launch (HIP events).
extern "C" int cfd_bench_gemm(cfd_handle c, int I, int J, int K, int tile_cfg, int iters, float* ms_out) {
  if (!c || !ms_out || K % 32 || I % CFD_D || I < CFD_D || J < 1 || iters < 1) return fail(CFD_E_ARG, "bad argument (I must be a multiple of 512)");
  HIPCHK(hipSetDevice(c->cfg.device));
  DBuf xf, yf, xs, ys, out;
  CHK(xf.ensure((size_t)I * K * 4));
  CHK(yf.ensure((size_t)J * K * 4));
  CHK(xs.ensure((size_t)I * K * 4));
  CHK(ys.ensure((size_t)J * K * 4));
  CHK(out.ensure((size_t)J * I * 4));
  HIPCHK(hipMemset(out.p, 0, (size_t)J * I * 4));
  long long n = (long long)I * K / 4;
  CHK(enqueue_philox_fill(xf.as<float>(), 1, I * K, 1ull, 0u, 0u, 3u, 0.05f, 0));
  n = (long long)J * K / 4;
  CHK(enqueue_philox_fill(yf.as<float>(), 1, (int)((long long)J * K), 2ull, 0u, 0u, 3u, 1.0f, 0));
  if (getenv("CFD_BENCH_ZERO")) {   // power/clock probe: all-zero operands
    HIPCHK(hipMemset(xf.p, 0, (size_t)I * K * 4));
    HIPCHK(hipMemset(yf.p, 0, (size_t)J * K * 4));
  }
  n = (long long)I * (K / 8);
  CHK(enqueue_to_split(c, CFD_PROF_OTHER, 0, xf.as<float>(), xs.as<char>(), (long long)I, K, (long long)K, (long long)K * 4, nullptr));
  n = (long long)J * (K / 8);
  CHK(enqueue_to_split(c, CFD_PROF_OTHER, 0, yf.as<float>(), ys.as<char>(), (long long)J, K, (long long)K, (long long)K * 4, nullptr));
  GemmArgs a = gemm_args();
  a.X[0] = xs.as<char>(); a.ldx[0] = (long long)K * 4; a.I[0] = I; a.Iclamp[0] = I; a.kt[0] = K / 32;
  a.Y = ys.as<char>(); a.ldy = (long long)K * 4; a.J = J; a.Jclamp = J;
  EpiResid e{out.as<float>(), 0, nullptr};
  EpiNull en{out.as<float>()};
  EpiF32 ef;
  memset(&ef, 0, sizeof(ef));
  ef.out = out.as<float>(); ef.ldo = I;
  const char* ev = getenv("CFD_BENCH_EPI");
  const int epi_kind = ev ? atoi(ev) : 0;   // 0 residual RMW, 1 no stores, 2 plain fp32 store, 3 split-pair store
  EpiSplit es;
  memset(&es, 0, sizeof(es));
  es.out = out.as<char>(); es.ldo = (long long)I * 4;
  if (epi_kind == 0 && I != CFD_D) return fail(CFD_E_ARG, "the residual epilogue has rows of 512");
  auto go = [&]() -> hipError_t {
    if (epi_kind == 1) return launch_gemm<MODE_PLAIN, EpiNull>(a, en, 1, 1, nullptr, tile_cfg);
    if (epi_kind == 2) return launch_gemm<MODE_PLAIN, EpiF32>(a, ef, 1, 1, nullptr, tile_cfg);
    if (epi_kind == 3) return launch_gemm<MODE_PLAIN, EpiSplit>(a, es, 1, 1, nullptr, tile_cfg);
    return launch_gemm<MODE_PLAIN, EpiResid>(a, e, 1, 1, nullptr, tile_cfg);
  };
  hipError_t err = go();   // warm-up
  if (err != hipSuccess) return fail(CFD_E_HIP, "gemm launch failed: %s", hipGetErrorString(err));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(c->pev[0], nullptr));
  for (int i = 0; i < iters; ++i) (void)go();
  HIPCHK(hipEventRecord(c->pev[1], nullptr));
  HIPCHK(hipEventSynchronize(c->pev[1]));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, c->pev[0], c->pev[1]));
  *ms_out = ms / iters;
  xf.release(); yf.release(); xs.release(); ys.release(); out.release();
  return CFD_OK;
}

