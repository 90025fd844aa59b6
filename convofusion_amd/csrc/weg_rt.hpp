// cfd_weg_eval on the row-tile kernels (rowtile.hpp forward with saved activations, rowtile_bwd.hpp backward): the product path of the
// word-excitation-guidance evaluation for small problems (the reference needs test batch size 1 for WEG, word_excitation_guidance.py:25).
// ~160 launches of ~5 us instead of weg_eval.hpp's ~400 launches of ~10 us, in the folded formulation the sampling loop itself uses.
// The evaluation has a workspace of its own (Ctx::wk[1]): it runs between two replays of an open sampling run's captured graph.
// Included by cfd_weg.hip after weg_eval.hpp.
#pragma once

namespace wegrt {

struct WorkGuard {   // the launches of an evaluation address Ctx::wk[1]; everything else of the handle keeps wk[0]
  Ctx* c;
  explicit WorkGuard(Ctx* c_) : c(c_) { c->w = &c->wk[1]; }
  ~WorkGuard() { c->w = &c->wk[0]; }
};

static bool eligible(Ctx* c, const cfd_weg_args* a) {
  if (!c->rt_on || !c->weg_rt_on || !c->hoist_memside || g_cfd_naive_gemm) return false;
  if (a->L > RT_MAX_L || (long long)a->B * a->L > c->rt_max_rows) return false;
  int sp = 0;
  for (int j = 0; j < CFD_NMEM; ++j) sp += (a->mem[j].S + 31) / 32 * 32;
  return sp <= RT_MAX_KEYS;
}

// (Re)build the evaluation's problem and arena when shapes or pointers change.  Host work only (allocations, row maps): never captured.
static int prepare(Ctx* c, const cfd_weg_args* a, int T, hipStream_t st) {
  WegRtState& s = c->wrt;
  const int B = a->B, L = a->L, nl = c->nl;
  std::vector<long long> sig = {B, L, nl, T};
  int spt = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {
    sig.push_back(a->mem[j].S);
    sig.push_back((long long)(size_t)a->mem[j].data);
    sig.push_back((long long)(size_t)a->mem[j].key_padding_mask);
    spt += (a->mem[j].S + 31) / 32 * 32;
  }
  if (sig == s.sig) return CFD_OK;
  HIPCHK(hipStreamSynchronize(st));
  s.sig.clear();
  const size_t M = (size_t)B * L, St = (size_t)a->mem[2].S;
  auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };   // floats, 256-byte granules
  size_t n = 0;
  const size_t n_x = al(M * CFD_D), n_qk = al(M * 2 * CFD_D), n_vt = al((size_t)B * CFD_D * RT_MAX_L), n_sc = al(M * spt), n_pre = al(M * CFD_FF);
  const size_t n_cst = al(M * 32 * 4);
  n += (size_t)(nl + 1) * 5 * n_x + (size_t)nl * (n_qk + n_vt + n_sc + n_pre + n_cst);
  const size_t n_att = al((size_t)B * nl * L * St), n_fws = al((size_t)B * (3 * (size_t)L * St + 3 * St + 64));
  n += 2 * n_att + n_fws + n_sc + 3 * n_x + 2 * n_x + n_pre + n_x + al(M * 3 * CFD_D);
  CHK(c->weg_rt_ws.ensure(n * 4));
  float* p = c->weg_rt_ws.as<float>();
  auto take = [&](size_t k) { float* r = p; p += k; return r; };
  for (int l = 0; l <= nl; ++l)
    for (int k = 0; k < 5; ++k) s.sv.x[l][k] = take(n_x);
  for (int l = 0; l < nl; ++l) {
    s.sv.qk[l] = reinterpret_cast<char*>(take(n_qk));
    s.sv.vt[l] = reinterpret_cast<char*>(take(n_vt));
    HIPCHK(hipMemset(s.sv.vt[l], 0, n_vt * 4));   // keys beyond L stay zero
    s.sv.sc[l] = take(n_sc);
    s.sv.cst[l] = take(n_cst);
    s.sv.pre[l] = take(n_pre);
  }
  s.att = take(n_att); s.d_att = take(n_att); s.fws = take(n_fws); s.dP = take(n_sc);
  for (int k = 0; k < 3; ++k) s.G[k] = take(n_x);
  s.dz = take(n_x); s.dy = take(n_x); s.dh = take(n_pre); s.dO = take(n_x); s.dqkv = take(al(M * 3 * CFD_D));
  {
    WorkGuard guard(c);
    cfd_memory mem[CFD_NMEM];
    float* att[CFD_NMEM] = {nullptr, nullptr, s.att, nullptr, nullptr};
    for (int j = 0; j < CFD_NMEM; ++j) mem[j] = a->mem[j];
    CHK(setup_problem(c, B, L, mem, att, 0, T));
    if (T > 1) {   // full tables: row t belongs to timestep t
      std::vector<int32_t> iota(T);
      for (int t = 0; t < T; ++t) iota[t] = t;
      HIPCHK(hipMemcpy(c->w->trows.p, iota.data(), (size_t)T * 4, hipMemcpyHostToDevice));
    }
    if (!c->w->pb.rt) return fail(CFD_E_STATE, "row-tile WEG evaluation: the problem does not qualify for the row-tile path");
  }
  s.sig = sig;
  return CFD_OK;
}

struct EvalArgs {
  const float* latents;        // dev [B][L][128] (staged)
  const int32_t *tok_off, *tok_idx;
  int last, nt_max;
  float k3[3];
  float *losses, *max_att, *grad;
};

template <int PRO, int EPI, int MAXSTEP>
static int bwd_gemm(Ctx* c, hipStream_t st, const RtBwdArgs& a, int N, int ntile) {
  const int lds = 16 * RT_BSTRIDE(a.K) * 4 + 8 * 1024;
  static unsigned long long attr = 0;
  if (!((attr >> (c->cfg.device & 63)) & 1ull)) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_bwd_gemm_kernel<PRO, EPI, MAXSTEP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               16 * RT_BSTRIDE(MAXSTEP * 32) * 4 + 8 * 1024));
    attr |= 1ull << (c->cfg.device & 63);
  }
  if (a.K != MAXSTEP * 32) return fail(CFD_E_ARG, "backward product: K = %d does not match the kernel instance (%d)", a.K, MAXSTEP * 32);
  hipLaunchKernelGGL((rt_bwd_gemm_kernel<PRO, EPI, MAXSTEP>), dim3(N / 16, ntile), dim3(512), lds, st, a);
  HIPCHK(hipGetLastError());
  ++c->wrt.launches;
  return CFD_OK;
}

// Launches only (capturable): [time tables + memory side when `full`], forward, objective, reverse sweep.
static int enqueue(Ctx* c, hipStream_t st, bool full, const EvalArgs& e) {
  WorkGuard guard(c);
  WegRtState& s = c->wrt;
  s.launches = 0;
  Work* w = c->w;
  const Problem& p = w->pb;
  const int nl = c->nl, B = p.Be, L = p.L, tpr = (L + 15) / 16, ntile = B * tpr, St = p.S[2];
  const long long M = p.M;
  if (full) {
    w->tt_key.clear();     // (this workspace's timestep-only tables are rebuilt from w->trows, whatever they held: cfd_problem.hip, build_time_tables)
    w->tt_mem_mask = 0;
    CHK(enqueue_time_tables(c, p.T, st));      // one row: the timestep index is in w->trows (copied in front of the launch sequence); full tables: row t = timestep t
    CHK(prepare_static_memside(c, st, 0, true));
    if (!p.rt) return fail(CFD_E_STATE, "row-tile WEG evaluation lost its path");
    s.launches += 20 + 5 * 6 + 2;
  }
  {
    const long long n = M * (CFD_LAT / 8);
    hipLaunchKernelGGL(to_split_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, e.latents, w->sample_sp.as<char>(), M, CFD_LAT,
                       (long long)CFD_LAT, (long long)CFD_LAT * 4, c->sat_in());
    HIPCHK(hipGetLastError());
  }
  CHK(enqueue_rows_rt(c, st, &s.sv));
  s.launches += 2 + 9 * nl - 3;
  if (L * (e.last - 1) <= WEG_SMALL_CELLS && e.nt_max <= WEG_SMALL_TOK && L <= 64)
    hipLaunchKernelGGL(weg_focus_small_kernel<>, dim3((unsigned)B), dim3(256), 0, st, s.att, e.tok_off, e.tok_idx, B, nl, L, St, e.last, e.k3[0], e.k3[1],
                       e.k3[2], e.losses, e.max_att, s.d_att);
  else
    hipLaunchKernelGGL(weg_focus_kernel<>, dim3((unsigned)B), dim3(256), 0, st, s.att, e.tok_off, e.tok_idx, B, nl, L, St, e.last, e.nt_max, e.k3[0], e.k3[1],
                       e.k3[2], s.fws, e.losses, e.max_att, s.d_att);
  HIPCHK(hipGetLastError());
  ++s.launches;

  // ---- reverse sweep (rowtile_bwd.hpp) -----------------------------------------------------------------------
  static unsigned long long attr = 0;
  const int lds_dp = 16 * RT_BSTRIDE(CFD_D) * 4 + 8 * 1024 + 64;
  const int lds_dy = 16 * RT_BSTRIDE(p.Sp_tot) * 4 + 8 * 1024 + 512 + 16 * 32 * 16 + 16 * 8 * 8;
  const int lds_sa = (4 * RT_MAX_L * (CFD_HD + 1) + 2 * RT_MAX_L * (RT_MAX_L + 1)) * 4;
  if (!((attr >> (c->cfg.device & 63)) & 1ull)) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_xbwd_dy_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               16 * RT_BSTRIDE(512) * 4 + 8 * 1024 + 512 + 16 * 32 * 16 + 16 * 8 * 8));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_xbwd_dy_kernel<RT_MAX_KEYS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               16 * RT_BSTRIDE(RT_MAX_KEYS) * 4 + 8 * 1024 + 512 + 16 * 32 * 16 + 16 * 8 * 8));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_selfattn_bwd_kernel<>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_sa));
    attr |= 1ull << (c->cfg.device & 63);
  }
  RtBwdArgs base;
  memset(&base, 0, sizeof(base));
  base.L = L; base.tpr = tpr;
  RtXBwdArgs xb;
  memset(&xb, 0, sizeof(xb));
  xb.L = L; xb.tpr = tpr; xb.nl = nl; xb.Sp_tot = p.Sp_tot;   // (this step's table rows: Work::now_*, set by the forward)
  xb.rsp = w->p_sp.as<float>(); xb.d_att = s.d_att; xb.dP = s.dP; xb.dy = s.dy;
  int nkb = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {
    xb.map[j] = p.map[j]; xb.S[j] = p.S[j]; xb.Sp[j] = p.Sp[j]; xb.off[j] = p.off[j];
    xb.blk0[j] = nkb; nkb += p.Sp[j] / 16;
    memcpy(xb.inst[j], p.rt_inst[j], RT_ARG_ROWS);
  }
  xb.blk0[CFD_NMEM] = nkb;
  xb.use_inst = p.rt_use_inst;
  auto Wraw = [&](int l, const char* name) -> const float* { return rawp(c, "decoder.layers." + std::to_string(l) + "." + name); };
  int gi = 0;                 // G[gi] holds the running gradient (valid once have_g)
  bool have_g = false;
  const float* dy1 = nullptr; // gradient at the next layer's norm1 output
  for (int l = nl - 1; l >= 0; --l) {
    const LayerW& lw = c->lw[l];
    RtXBwdArgs x5 = xb;
    x5.layer = l; x5.sc = s.sv.sc[l]; x5.cst = s.sv.cst[l];
    for (int j = 0; j < CFD_NMEM; ++j) {
      const size_t rows = (size_t)p.U[j] * p.Sp[j];
      x5.K[j] = w->kall_sp[j].as<char>() + (size_t)l * rows * CFD_D * 4;
      x5.VT[j] = w->vt_all[j].as<char>() + (size_t)l * rows * CFD_D * 4;
      x5.kb[j] = w->now_kb[j] + (size_t)l * CFD_D;
      x5.vb[j] = w->now_vb[j] + (size_t)l * CFD_D;
    }
    if (l < nl - 1) {
      const LayerW& up = c->lw[l + 1];
      {   // B1: through the next layer's norm1 into this layer's output, then linear2 and the GELU
        RtBwdArgs a = base;
        a.K = CFD_D; a.a = dy1; a.g = s.G[gi]; a.x = s.sv.x[l + 1][0]; a.gamma = up.ln1g; a.gout = s.G[(gi + 1) % 3];
        a.w = Wraw(l, "linear2.weight"); a.ldw = CFD_FF; a.out = s.dh; a.ldo = CFD_FF; a.pre = s.sv.pre[l];
        CHK((bwd_gemm<RT_BPRO_LN, RT_BEPI_GELU, 16>(c, st, a, CFD_FF, ntile)));
        gi = (gi + 1) % 3;
      }
      {   // B2: linear1
        RtBwdArgs a = base;
        a.K = CFD_FF; a.a = s.dh; a.w = Wraw(l, "linear1.weight"); a.ldw = CFD_D; a.out = s.dy; a.ldo = CFD_D;
        CHK((bwd_gemm<RT_BPRO_ROWS, RT_BEPI_F32, 32>(c, st, a, CFD_D, ntile)));
      }
      {   // B3: norm3, then time block 2's projection
        RtBwdArgs a = base;
        a.K = CFD_D; a.a = s.dy; a.g = s.G[gi]; a.x = s.sv.x[l][4]; a.gamma = lw.ln3g; a.gout = s.G[(gi + 1) % 3];
        a.w = Wraw(l, "time_block2.out_layers.2.weight"); a.ldw = CFD_D; a.out = s.dz; a.ldo = CFD_D;
        CHK((bwd_gemm<RT_BPRO_LN, RT_BEPI_F32, 16>(c, st, a, CFD_D, ntile)));
        gi = (gi + 1) % 3;
      }
      {   // B4: time block 2's SiLU / modulation / norm, then the probabilities' gradient
        RtXBwdArgs a = x5;
        a.dz = s.dz; a.g = s.G[gi]; a.x = s.sv.x[l][3]; a.gamma = lw.tb2g; a.beta = lw.tb2b;
        a.ss = w->now_ss + (size_t)(2 * l + 1) * 2 * CFD_D; a.gout = s.G[(gi + 1) % 3];
        hipLaunchKernelGGL(rt_xbwd_dp_kernel<>, dim3(nkb, ntile), dim3(512), lds_dp, st, a);
        HIPCHK(hipGetLastError());
        ++s.launches;
        gi = (gi + 1) % 3;
      }
    }
    {   // B5: softmax backward and the folded keys
      RtXBwdArgs a = x5;
      a.dp_from_datt = have_g || l < nl - 1 ? 0 : 1;
      if (p.Sp_tot <= 512) hipLaunchKernelGGL(rt_xbwd_dy_kernel<512>, dim3(CFD_D / 16, ntile), dim3(512), lds_dy, st, a);
      else hipLaunchKernelGGL(rt_xbwd_dy_kernel<RT_MAX_KEYS>, dim3(CFD_D / 16, ntile), dim3(512), lds_dy, st, a);
      HIPCHK(hipGetLastError());
      ++s.launches;
    }
    {   // B6: norm2, then time block 1's projection
      RtBwdArgs a = base;
      a.K = CFD_D; a.a = s.dy; a.g = (l < nl - 1) ? s.G[gi] : nullptr; a.x = s.sv.x[l][2]; a.gamma = lw.ln2g; a.gout = s.G[(gi + 1) % 3];
      a.w = Wraw(l, "time_block1.out_layers.2.weight"); a.ldw = CFD_D; a.out = s.dz; a.ldo = CFD_D;
      CHK((bwd_gemm<RT_BPRO_LN, RT_BEPI_F32, 16>(c, st, a, CFD_D, ntile)));
      gi = (gi + 1) % 3;
      have_g = true;
    }
    {   // B7: time block 1, then the attention's output projection
      RtBwdArgs a = base;
      a.K = CFD_D; a.a = s.dz; a.g = s.G[gi]; a.x = s.sv.x[l][1]; a.gamma = lw.tb1g; a.beta = lw.tb1b;
      a.ss = w->now_ss + (size_t)(2 * l) * 2 * CFD_D; a.gout = s.G[(gi + 1) % 3];
      a.w = Wraw(l, "self_attn.out_proj.weight"); a.ldw = CFD_D; a.out = s.dO; a.ldo = CFD_D;
      CHK((bwd_gemm<RT_BPRO_TB, RT_BEPI_F32, 16>(c, st, a, CFD_D, ntile)));
      gi = (gi + 1) % 3;
    }
    {   // B8: attention core
      RtSelfBwdArgs a{s.sv.qk[l], s.sv.vt[l], s.dO, s.dqkv, L, (float)std::sqrt(1.0 / (double)CFD_HD)};
      hipLaunchKernelGGL(rt_selfattn_bwd_kernel<>, dim3(CFD_NHEAD, B), dim3(256), lds_sa, st, a);
      HIPCHK(hipGetLastError());
      ++s.launches;
    }
    {   // B9: packed in-projection
      RtBwdArgs a = base;
      a.K = 3 * CFD_D; a.a = s.dqkv; a.w = Wraw(l, "self_attn.in_proj_weight"); a.ldw = CFD_D; a.out = s.dy; a.ldo = CFD_D;
      CHK((bwd_gemm<RT_BPRO_ROWS, RT_BEPI_F32, 48>(c, st, a, CFD_D, ntile)));
      dy1 = s.dy;
    }
  }
  {   // through layer 0's norm1 and the latent embedding
    RtBwdArgs a = base;
    a.K = CFD_D; a.a = dy1; a.g = s.G[gi]; a.x = s.sv.x[0][0]; a.gamma = c->lw[0].ln1g; a.gout = s.G[(gi + 1) % 3];
    a.w = rawp(c, "latent_embd.weight"); a.ldw = CFD_LAT; a.out = e.grad; a.ldo = CFD_LAT;
    CHK((bwd_gemm<RT_BPRO_LN, RT_BEPI_F32, 16>(c, st, a, CFD_LAT, ntile)));
  }
  return CFD_OK;
}

}  // namespace wegrt
