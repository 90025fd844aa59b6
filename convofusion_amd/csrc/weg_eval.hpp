// cfd_weg_eval: one evaluation of the word-excitation-guidance objective on the text-only chunk and its gradient with
// respect to the latents, enqueued from C++ (no per-launch host language overhead).  What the reference does with
//   latents.requires_grad_(True); _, att = denoiser(latents, t, text_only_states, ...); loss = focus(att[2]);
//   torch.autograd.grad(loss, latents)                       (convofusion.py:447-471,490-495; weg.py:11-81)
// Forward in the reference's own float32 formulation (denoiser.py:173-386, cross_attention.py:556-664, 426-439) keeping
// the activations the backward needs, the objective (weg_focus_kernel), then the reverse sweep; only the query side
// carries gradient (memories, time embedding, weights are constants).  Batch-major rows (r = b * L + l) so latents,
// memories and the gradient need no permutation; every head split / transpose is a strided view of gemm_f32_kernel.
// Included by cfd_weg.hip after the handle definition (cfd_internal.hpp).
#pragma once

namespace weg {

struct View {           // element (z1, z2, r, c) = p[z1*b1 + z2*b2 + r*rs + c*cs]
  float* p;
  long long rs, cs, b1, b2;
};

struct AttnSaved {
  float *q, *k, *v, *p;      // q rows q_rs apart ([B][T] rows), k / v rows kv_rs apart ([B][S] rows); probabilities of (b, h) at p + b*pb1 + h*T*S
  long long q_rs, kv_rs, pb1;
  const float *W, *Wo;       // in_proj_weight [3E][E], out_proj.weight [E][E]
  int T, S, H;
  float scale;
};

struct TbSaved {
  float *x, *h, *e;
  const float *g, *Wout;
};

struct LayerSaved {
  float *x0, *x2, *x4, *ffn_pre;
  AttnSaved self, cross[CFD_NMEM];
  TbSaved tb1, tb2;
};

struct Ctx {
  cfd_handle c;
  hipStream_t st;
  bool dry;            // sizing pass: count workspace bytes, launch nothing
  char* base;
  size_t off;
  int B, L, D, E;
  int err;
  std::string missing;
  int launches;
  bool reuse = false;      // the memory-side / time-only results of the previous evaluation are still in the arena: skip their launches
  bool suppress = false;

  float* alloc(size_t n) {
    const size_t bytes = (n * 4 + 255) & ~(size_t)255;
    float* p = reinterpret_cast<float*>(base + off);   // sizing pass: a fake non-null base, never dereferenced (same control flow)
    off += bytes;
    return p;
  }
  const float* W(const std::string& name) {
    auto it = c->raw.find(name);
    if (it == c->raw.end()) {
      if (missing.empty()) missing = name;
      err = CFD_E_STATE;
      return nullptr;
    }
    return it->second.as<float>();
  }
  bool skip() {
    if (suppress) return true;
    ++launches;
    return dry || err;
  }
  // launches between hold(true) and hold(false) depend on the timestep and the memories only
  void hold(bool on) { suppress = on && reuse; }
  void gemm(int M, int N, int K, int nb1, int nb2, View A, View Bv, View Cv, const float* bias, float alpha, int accumulate,
            const float* resid = nullptr, int a_act = 0) {
    if (skip()) return;
    MatView a{A.p, A.rs, A.cs, A.b1, A.b2}, b{Bv.p, Bv.rs, Bv.cs, Bv.b1, Bv.b2};
    launch_gemm_f32(st, a, b, Cv.p, Cv.rs, Cv.cs, Cv.b1, Cv.b2, M, N, K, nb1, nb2, bias, alpha, accumulate, resid, a_act);
  }
  // F.linear: x rows x_rs apart, [rows][K] -> out rows out_rs apart [rows][N]; w [N][K]; + resid (out's layout); a_act on x
  float* linear(const float* x, long long x_rs, long long rows, int K, const float* w, const float* bias, int N, float* out = nullptr,
                long long out_rs = 0, const float* resid = nullptr, int a_act = 0) {
    if (!out) {
      out = alloc((size_t)rows * N);
      out_rs = N;
    }
    gemm((int)rows, N, K, 1, 1, View{const_cast<float*>(x), x_rs, 1, 0, 0}, View{const_cast<float*>(w), 1, K, 0, 0}, View{out, out_rs, 1, 0, 0}, bias,
         1.0f, 0, resid, a_act);
    return out;
  }
  // gradient of F.linear(x, w) with respect to x: dy (rows dy_rs apart) [rows][N] @ w [N][K]
  float* linear_bwd(const float* dy, long long dy_rs, long long rows, int N, const float* w, int K, float* out = nullptr, int accumulate = 0) {
    if (!out) out = alloc((size_t)rows * K);
    gemm((int)rows, K, N, 1, 1, View{const_cast<float*>(dy), dy_rs, 1, 0, 0}, View{const_cast<float*>(w), K, 1, 0, 0}, View{out, K, 1, 0, 0}, nullptr,
         1.0f, accumulate);
    return out;
  }
  void softmax(float* s, long long rows, int Lk, const uint8_t* kpm, long long rows_per_batch) {
    if (skip()) return;
    hipLaunchKernelGGL(softmax_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, s, kpm, rows, Lk, rows_per_batch);
  }
  void softmax_bwd(const float* p, float* dp, const float* extra, long long rows, int Lk) {
    if (skip()) return;
    hipLaunchKernelGGL(softmax_bwd_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, dp, extra, rows, Lk);
  }
  float* ln(const float* x, long long rows, const float* g, const float* b) {
    float* out = alloc((size_t)rows * D);
    if (!skip()) hipLaunchKernelGGL(layernorm_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, g, b, out, rows, D, 1e-5f);
    return out;
  }
  void ln_bwd(const float* x, const float* g, const float* dy, float* dx, long long rows, int accumulate, const float* tb_h = nullptr,
              const float* tb_e = nullptr) {
    if (skip()) return;
    hipLaunchKernelGGL(layernorm_bwd_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, g, dy, dx, rows, D, 1e-5f, accumulate, tb_h, tb_e);
  }
  float* ln_mod(const float* x, long long rows, const float* g, const float* b, const float* e) {
    float* out = alloc((size_t)rows * D);
    if (!skip()) hipLaunchKernelGGL(layernorm_mod_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, g, b, e, out, rows, D, 1e-5f);
    return out;
  }
  void gemm_sum(const GemmSum& gs) {
    if (skip()) return;
    launch_gemm_f32_sum(st, gs);
  }
  void gemm_grouped(GemmGroups& gs) {
    if (skip()) return;
    launch_gemm_f32_grouped(st, gs);
  }
  void ln_grouped(const LnGroups& gs, int n) {
    if (skip()) return;
    long long mx = 0;
    for (int i = 0; i < n; ++i) mx = std::max(mx, gs.rows[i]);
    hipLaunchKernelGGL(layernorm_f32_grouped_kernel<>, dim3((unsigned)((mx + 3) / 4), (unsigned)n), dim3(256), 0, st, gs, D, 1e-5f);
  }
  void softmax_grouped(const SoftmaxGroups& gs, int n, bool backward) {
    if (skip()) return;
    long long mx = 0;
    for (int i = 0; i < n; ++i) mx = std::max(mx, gs.rows[i]);
    if (backward) hipLaunchKernelGGL(softmax_bwd_f32_grouped_kernel<>, dim3((unsigned)((mx + 3) / 4), (unsigned)n), dim3(256), 0, st, gs);
    else hipLaunchKernelGGL(softmax_f32_grouped_kernel<>, dim3((unsigned)((mx + 3) / 4), (unsigned)n), dim3(256), 0, st, gs);
  }
  void ew(int op, const float* a, const float* b, float* out, long long n, int Dd = 1, int R1 = 1, long long s0 = 0, long long s1 = 0, float alpha = 0.f) {
    if (skip()) return;
    hipLaunchKernelGGL(ew_f32_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op, a, b, out, n, Dd, R1, s0, s1, alpha);
  }
};

// head views of batch-major rows (row stride rs, n rows per batch entry): (b, h; t, d) and its transpose (b, h; d, t)
static inline View heads(float* t, int n, long long rs, int hd) { return View{t, rs, 1, (long long)n * rs, hd}; }
static inline View heads_T(float* t, int n, long long rs, int hd) { return View{t, 1, rs, (long long)n * rs, hd}; }

static inline GemmGroup group(View A, View Bv, View Cv, int M, int N, int K, int nb1, int nb2, const float* bias, float alpha);

// nn.MultiheadAttention(query, memory, memory, key_padding_mask) on batch-major rows; self-attention (memory == query) projects
// q | k | v in one product, cross-attention k | v.  The output projection writes rows out_rs apart (a column block of the
// concatenated tensor, cross_attention.py:629) and adds `resid` (the residual connection) when given.
static float* mha_fwd(Ctx& x, const std::string& pfx, const float* query, int T, const float* memory, int S, int H, const uint8_t* kpm, bool self,
                      bool use_p_out, float* p_out, long long pb1, float* out, long long out_rs, const float* resid, AttnSaved& sv) {
  const int E = x.E, hd = E / H, B = x.B;
  const float* W = x.W(pfx + ".in_proj_weight");
  const float* Bi = x.W(pfx + ".in_proj_bias");
  const float* Wo = x.W(pfx + ".out_proj.weight");
  const float* bo = x.W(pfx + ".out_proj.bias");
  if (x.err) return nullptr;
  float *q, *k, *v;
  long long q_rs, kv_rs;
  if (self) {
    q = x.linear(query, E, (long long)B * T, E, W, Bi, 3 * E);
    k = q + E;
    v = q + 2 * E;
    q_rs = kv_rs = 3 * E;
  } else {
    q = x.linear(query, E, (long long)B * T, E, W, Bi, E);
    k = x.linear(memory, E, (long long)B * S, E, W + (size_t)E * E, Bi + E, 2 * E);
    v = k + E;
    q_rs = E;
    kv_rs = 2 * E;
  }
  float* p = p_out;
  if (!use_p_out) {
    p = x.alloc((size_t)B * H * T * S);
    pb1 = (long long)H * T * S;
  }
  const float scale = (float)std::sqrt(1.0 / (double)hd);
  x.gemm(T, S, hd, B, H, heads(q, T, q_rs, hd), heads_T(k, S, kv_rs, hd), View{p, S, 1, pb1, (long long)T * S}, nullptr, scale, 0);
  if (pb1 == (long long)H * T * S) x.softmax(p, (long long)B * H * T, S, kpm, (long long)H * T);
  else
    for (int b = 0; b < B; ++b) x.softmax(p + b * pb1, (long long)H * T, S, kpm ? kpm + (size_t)b * S : nullptr, (long long)H * T);
  float* o = x.alloc((size_t)B * T * E);
  x.gemm(T, hd, S, B, H, View{p, S, 1, pb1, (long long)T * S}, heads(v, S, kv_rs, hd), heads(o, T, E, hd), nullptr, 1.0f, 0);
  out = x.linear(o, E, (long long)B * T, E, Wo, bo, E, out, out_rs, resid);
  sv = AttnSaved{q, k, v, p, q_rs, kv_rs, pb1, W, Wo, T, S, H, scale};
  return out;
}

// Gradient with respect to the query input into dx (+= when accumulate); self-attention: the query / key / value paths in
// one product against the packed in-projection.  `dout` (rows dout_rs apart) may be null: nothing arrives through the output;
// `d_prob` (layout of p, blocks dpb1 apart) arrives at the probabilities.
static void mha_bwd(Ctx& x, const AttnSaved& sv, const float* dout, long long dout_rs, const float* d_prob, long long dpb1, bool self, float* dx,
                    int accumulate) {
  const int E = x.E, H = sv.H, hd = E / H, B = x.B, T = sv.T, S = sv.S;
  const long long blk = (long long)H * T * S;
  float* dp = x.alloc((size_t)B * blk);
  float* d_o = nullptr;
  if (dout) {
    d_o = x.linear_bwd(dout, dout_rs, (long long)B * T, E, sv.Wo, E);
    x.gemm(T, S, hd, B, H, heads(d_o, T, E, hd), heads_T(sv.v, S, sv.kv_rs, hd), View{dp, S, 1, blk, (long long)T * S}, nullptr, 1.0f, 0);
  }
  // without dout the gradient at the probabilities is d_prob alone: dp is read as zeros through the `extra`-only form below
  if (!dout) {
    if (!x.skip()) (void)hipMemsetAsync(dp, 0, (size_t)B * blk * 4, x.st);
  }
  if (sv.pb1 == blk && (!d_prob || dpb1 == blk)) x.softmax_bwd(sv.p, dp, d_prob, (long long)B * H * T, S);
  else
    for (int b = 0; b < B; ++b) x.softmax_bwd(sv.p + b * sv.pb1, dp + b * blk, d_prob ? d_prob + b * dpb1 : nullptr, (long long)H * T, S);
  if (!self) {
    float* dq = x.alloc((size_t)B * T * E);
    x.gemm(T, hd, S, B, H, View{dp, S, 1, blk, (long long)T * S}, heads(sv.k, S, sv.kv_rs, hd), heads(dq, T, E, hd), nullptr, sv.scale, 0);
    x.linear_bwd(dq, E, (long long)B * T, E, sv.W, E, dx, accumulate);
    return;
  }
  float* dqkv = x.alloc((size_t)B * T * 3 * E);      // [rows][dq | dk | dv]: three products, one launch
  GemmGroups g3;
  g3.n = 3;
  g3.g[0] = group(View{dp, S, 1, blk, (long long)T * S}, heads(sv.k, S, sv.kv_rs, hd), heads(dqkv, T, 3 * E, hd), T, hd, S, B, H, nullptr, sv.scale);
  g3.g[1] = group(View{dp, 1, S, blk, (long long)T * S}, heads(sv.q, T, sv.q_rs, hd), heads(dqkv + E, S, 3 * E, hd), S, hd, T, B, H, nullptr, sv.scale);
  g3.g[2] = group(View{sv.p, 1, S, sv.pb1, (long long)T * S}, heads(d_o, T, E, hd), heads(dqkv + 2 * E, S, 3 * E, hd), S, hd, T, B, H, nullptr, 1.0f);
  x.gemm_grouped(g3);
  x.linear_bwd(dqkv, 3 * E, (long long)B * T, 3 * E, sv.W, E, dx, accumulate);
}

static inline GemmGroup group(View A, View Bv, View Cv, int M, int N, int K, int nb1, int nb2, const float* bias, float alpha) {
  GemmGroup g;
  g.A = MatView{A.p, A.rs, A.cs, A.b1, A.b2};
  g.B = MatView{Bv.p, Bv.rs, Bv.cs, Bv.b1, Bv.b2};
  g.C = Cv.p; g.c_rs = Cv.rs; g.c_cs = Cv.cs; g.c_b1 = Cv.b1; g.c_b2 = Cv.b2;
  g.M = M; g.N = N; g.K = K; g.nb1 = nb1; g.nb2 = nb2;
  g.bias = bias; g.resid = nullptr; g.alpha = alpha; g.accumulate = 0; g.a_act = 0;
  return g;
}

// The five single-head cross-attentions of a layer (cross_attention.py:581-629) with one launch per stage instead of five:
// memory LayerNorms, query projections, key|value projections, scores, softmax, P.V, output projections into the column
// blocks of `cat`.  The tlsn probabilities go to `att_i` (blocks att_b1 apart: Denoiser.forward's att_mats[2] of this layer).
static void cross_fwd_grouped(Ctx& x, const std::string& p, const float* t2, float* const mems[CFD_NMEM], const cfd_memory* mem, float* att_i,
                              long long att_b1, float* cat, AttnSaved sv[CFD_NMEM]) {
  const int E = x.E, B = x.B, T = x.L;
  const long long rows = (long long)B * T;
  const float *W[CFD_NMEM], *Bi[CFD_NMEM], *Wo[CFD_NMEM], *bo[CFD_NMEM];
  LnGroups ln;
  float *mn[CFD_NMEM], *q[CFD_NMEM], *kv[CFD_NMEM], *pr[CFD_NMEM], *o[CFD_NMEM];
  long long pb1[CFD_NMEM];
  for (int j = 0; j < CFD_NMEM; ++j) {
    const std::string nm = MEM_NAMES[j], a = p + "multihead_attn_" + nm;
    W[j] = x.W(a + ".in_proj_weight"); Bi[j] = x.W(a + ".in_proj_bias"); Wo[j] = x.W(a + ".out_proj.weight"); bo[j] = x.W(a + ".out_proj.bias");
    const int S = mem[j].S;
    mn[j] = x.alloc((size_t)B * S * E);
    ln.x[j] = mems[j]; ln.g[j] = x.W(p + nm + "_norm.weight"); ln.b[j] = x.W(p + nm + "_norm.bias"); ln.out[j] = mn[j]; ln.rows[j] = (long long)B * S;
    q[j] = x.alloc((size_t)rows * E);
    kv[j] = x.alloc((size_t)B * S * 2 * E);
    if (j == 2) { pr[j] = att_i; pb1[j] = att_b1; }
    else { pr[j] = x.alloc((size_t)B * T * S); pb1[j] = (long long)T * S; }
    o[j] = x.alloc((size_t)rows * E);
  }
  if (x.err) return;
  x.hold(true);
  x.ln_grouped(ln, CFD_NMEM);
  x.hold(false);
  const float scale = (float)std::sqrt(1.0 / (double)E);
  GemmGroups gq, gkv, gs, gpv, go;
  SoftmaxGroups sm;
  gq.n = gkv.n = gs.n = gpv.n = go.n = CFD_NMEM;
  for (int j = 0; j < CFD_NMEM; ++j) {
    const int S = mem[j].S;
    gq.g[j] = group(View{const_cast<float*>(t2), E, 1, 0, 0}, View{const_cast<float*>(W[j]), 1, E, 0, 0}, View{q[j], E, 1, 0, 0}, (int)rows, E, E, 1, 1, Bi[j], 1.0f);
    gkv.g[j] = group(View{mn[j], E, 1, 0, 0}, View{const_cast<float*>(W[j]) + (size_t)E * E, 1, E, 0, 0}, View{kv[j], 2 * E, 1, 0, 0}, B * S, 2 * E, E, 1, 1,
                     Bi[j] + E, 1.0f);
    gs.g[j] = group(heads(q[j], T, E, E), heads_T(kv[j], S, 2 * E, E), View{pr[j], S, 1, pb1[j], (long long)T * S}, T, S, E, B, 1, nullptr, scale);
    sm.s[j] = pr[j]; sm.p[j] = nullptr; sm.extra[j] = nullptr; sm.kpm[j] = mem[j].key_padding_mask; sm.rows[j] = rows; sm.rows_per_batch[j] = T;
    sm.s_bstride[j] = pb1[j]; sm.p_bstride[j] = 0; sm.e_bstride[j] = 0; sm.Lk[j] = S;
    gpv.g[j] = group(View{pr[j], S, 1, pb1[j], (long long)T * S}, heads(kv[j] + E, S, 2 * E, E), heads(o[j], T, E, E), T, E, S, B, 1, nullptr, 1.0f);
    go.g[j] = group(View{o[j], E, 1, 0, 0}, View{const_cast<float*>(Wo[j]), 1, E, 0, 0}, View{cat + (size_t)j * E, (long long)CFD_NMEM * E, 1, 0, 0}, (int)rows, E, E,
                    1, 1, bo[j], 1.0f);
    sv[j] = AttnSaved{q[j], kv[j], kv[j] + E, pr[j], E, 2 * E, pb1[j], W[j], Wo[j], T, S, 1, scale};
  }
  x.gemm_grouped(gq);
  x.hold(true);
  x.gemm_grouped(gkv);
  x.hold(false);
  x.gemm_grouped(gs);
  x.softmax_grouped(sm, CFD_NMEM, false);
  x.gemm_grouped(gpv);
  x.gemm_grouped(go);
}

// Their backward with respect to the shared query input: dt2 = sum_j dq_j Wq_j.  `dcat` [rows][5 E] arrives through the
// outputs, `d_att_i` (blocks d_b1 apart) at the tlsn probabilities.
static void cross_bwd_grouped(Ctx& x, const AttnSaved sv[CFD_NMEM], const float* dcat, const float* d_att_i, long long d_b1, float* dt2) {
  const int E = x.E, B = x.B, T = x.L;
  const long long rows = (long long)B * T;
  float *d_o[CFD_NMEM], *dp[CFD_NMEM], *dq[CFD_NMEM];
  GemmGroups g1, g2, g3;
  SoftmaxGroups sm;
  g1.n = g2.n = g3.n = CFD_NMEM;
  for (int j = 0; j < CFD_NMEM; ++j) {
    const int S = sv[j].S;
    const long long blk = (long long)T * S;
    d_o[j] = x.alloc((size_t)rows * E);
    dp[j] = x.alloc((size_t)B * blk);
    dq[j] = x.alloc((size_t)rows * E);
    g1.g[j] = group(View{const_cast<float*>(dcat) + (size_t)j * E, (long long)CFD_NMEM * E, 1, 0, 0}, View{const_cast<float*>(sv[j].Wo), E, 1, 0, 0},
                    View{d_o[j], E, 1, 0, 0}, (int)rows, E, E, 1, 1, nullptr, 1.0f);
    g2.g[j] = group(heads(d_o[j], T, E, E), heads_T(sv[j].v, S, sv[j].kv_rs, E), View{dp[j], S, 1, blk, blk}, T, S, E, B, 1, nullptr, 1.0f);
    sm.s[j] = dp[j]; sm.p[j] = sv[j].p; sm.extra[j] = j == 2 ? d_att_i : nullptr; sm.kpm[j] = nullptr; sm.rows[j] = rows; sm.rows_per_batch[j] = T;
    sm.s_bstride[j] = blk; sm.p_bstride[j] = sv[j].pb1; sm.e_bstride[j] = d_b1; sm.Lk[j] = S;
    g3.g[j] = group(View{dp[j], S, 1, blk, blk}, heads(sv[j].k, S, sv[j].kv_rs, E), heads(dq[j], T, E, E), T, E, S, B, 1, nullptr, sv[j].scale);
  }
  x.gemm_grouped(g1);
  x.gemm_grouped(g2);
  x.softmax_grouped(sm, CFD_NMEM, true);
  x.gemm_grouped(g3);
  GemmSum sum;                                                   // dt2 = sum_j dq_j Wq_j
  sum.g = group(View{dq[0], E, 1, 0, 0}, View{const_cast<float*>(sv[0].W), E, 1, 0, 0}, View{dt2, E, 1, 0, 0}, (int)rows, E, E, 1, 1, nullptr, 1.0f);
  sum.n_more = CFD_NMEM - 1;
  for (int j = 1; j < CFD_NMEM; ++j)
    sum.more[j - 1] = GemmSeg{MatView{dq[j], E, 1, 0, 0}, MatView{sv[j].W, E, 1, 0, 0}, E};
  x.gemm_sum(sum);
}

// x + TimeBlock(x) (cross_attention.py:426-439, the caller's residual :575,:655); `e` = emb_layers(temb) [1][2 D] (scale first),
// computed for all blocks of the evaluation at once (time_block_embeddings)
static float* time_block_fwd(Ctx& x, const std::string& pfx, float* in, float* e, TbSaved& sv) {
  const int D = x.D;
  const long long rows = (long long)x.B * x.L;
  const float *g = x.W(pfx + ".norm.weight"), *bn = x.W(pfx + ".norm.bias"), *Wout = x.W(pfx + ".out_layers.2.weight"), *bout = x.W(pfx + ".out_layers.2.bias");
  if (x.err) return nullptr;
  float* h = x.ln_mod(in, rows, g, bn, e);                                            // LN(x) (1 + scale) + shift
  sv = TbSaved{in, h, e, g, Wout};
  return x.linear(h, D, rows, D, Wout, bout, D, nullptr, 0, in, 1);                   // in + Linear(SiLU(h))
}

// g += d TimeBlock(x)/dx applied to g
static void time_block_bwd(Ctx& x, const TbSaved& sv, float* g) {
  const int D = x.D;
  const long long rows = (long long)x.B * x.L;
  float* dh = x.linear_bwd(g, D, rows, D, sv.Wout, D);
  x.ln_bwd(sv.x, sv.g, dh, g, rows, 1, sv.h, sv.e);                                   // through SiLU', the modulation and the norm
}

// emb_layers of every TimeBlock of the evaluation: Linear(SiLU(temb)) -> e[2 * layer + {0, 1}] [1][2 D], five blocks per launch
static void time_block_embeddings(Ctx& x, const float* temb, int n_layers, std::vector<float*>& e) {
  const int D = x.D;
  e.assign((size_t)2 * n_layers, nullptr);
  GemmGroups gs;
  gs.n = 0;
  for (int i = 0; i < 2 * n_layers; ++i) {
    const std::string pfx = "decoder.layers." + std::to_string(i / 2) + (i % 2 ? ".time_block2" : ".time_block1");
    const float *We = x.W(pfx + ".emb_layers.1.weight"), *be = x.W(pfx + ".emb_layers.1.bias");
    if (x.err) return;
    e[i] = x.alloc((size_t)2 * D);
    gs.g[gs.n] = group(View{const_cast<float*>(temb), D, 1, 0, 0}, View{const_cast<float*>(We), 1, D, 0, 0}, View{e[i], 2 * D, 1, 0, 0}, 1, 2 * D, D, 1, 1, be, 1.0f);
    gs.g[gs.n].a_act = 1;
    if (++gs.n == GEMM_MAX_GROUPS || i == 2 * n_layers - 1) {
      x.gemm_grouped(gs);
      gs.n = 0;
    }
  }
}

struct Args {
  const float* latents;        // dev [B][L][latent]
  const float* trow;           // dev: the timestep's row of the sinusoid table (get_timestep_embedding)
  const cfd_memory* mem;       // 5 memories, U == B, batch-major [B][S][D]
  const int32_t *tok_off, *tok_idx;   // dev
  int last, nt_max;
  float k3[3];
  float *losses, *max_att, *grad;    // dev outputs
};

// One pass over the whole evaluation; with x.dry it only sizes the workspace.
static void run(Ctx& x, const Args& a) {
  cfd_handle c = x.c;
  const int B = x.B, L = x.L, D = x.D, NL = c->nl, LAT = c->cfg.latent_dim, FF = c->cfg.ff_size;
  const long long rows = (long long)B * L;
  // ---- embedding, time embedding, memories (denoiser.py:183-353)
  float* xx = x.linear(a.latents, LAT, rows, LAT, x.W("latent_embd.weight"), x.W("latent_embd.bias"), D);
  x.hold(true);
  float* t1 = x.linear(a.trow, D, 1, D, x.W("time_embedding.linear_1.weight"), x.W("time_embedding.linear_1.bias"), D);
  float* temb = x.linear(t1, D, 1, D, x.W("time_embedding.linear_2.weight"), x.W("time_embedding.linear_2.bias"), D, nullptr, 0, nullptr, 1);
  x.hold(false);
  x.ew(EW_ADD_BCAST, xx, x.W("bh_embedding.weight"), xx, rows * D, D, 2, 0, D);                 // token l gets bh[l % 2] (:316-317)
  const float* qpe = x.W("query_pos.pe");
  for (int b = 0; b < B; ++b)                                                                      // and pe[l / 2] (SineBH)
    x.ew(EW_ADD_BCAST, xx + (size_t)b * L * D, qpe, xx + (size_t)b * L * D, (long long)L * D, D, 2, D, 0);
  float* mems[CFD_NMEM];
  const float *ce = x.W("condition_embedding.weight"), *mpe = x.W("mem_pos.pe");
  x.hold(true);
  for (int j = 0; j < CFD_NMEM; ++j) {
    const int S = a.mem[j].S;
    const long long n = (long long)B * S * D;
    mems[j] = x.alloc((size_t)n);
    x.ew(EW_ADD_BCAST, a.mem[j].data, temb, mems[j], n, D, S, 0, 0);                            // + temb (:223-261)
    x.ew(EW_ADD_BCAST, mems[j], ce ? ce + (size_t)j * D : nullptr, mems[j], n, D, S, 0, 0);       // + condition id (:332-353)
    x.ew(EW_ADD_BCAST, mems[j], mpe, mems[j], n, D, S, 0, D);                                   // + pe[s]
  }
  if (x.err) return;
  std::vector<float*> tb_e;
  time_block_embeddings(x, temb, NL, tb_e);
  x.hold(false);
  if (x.err) return;
  // ---- layers, keeping what the backward needs
  const int St = a.mem[2].S;
  float* att = x.alloc((size_t)B * NL * L * St);                 // [B][NL][L][S_text]: Denoiser.forward's att_mats[2]
  std::vector<LayerSaved> sv(NL);
  for (int i = 0; i < NL; ++i) {
    const std::string p = "decoder.layers." + std::to_string(i) + ".";
    LayerSaved& s = sv[i];
    s.x0 = xx;
    float* t2 = x.ln(xx, rows, x.W(p + "norm1.weight"), x.W(p + "norm1.bias"));
    xx = mha_fwd(x, p + "self_attn", t2, L, t2, L, c->cfg.num_heads, nullptr, true, false, nullptr, 0, nullptr, 0, xx, s.self);
    if (x.err) return;
    xx = time_block_fwd(x, p + "time_block1", xx, tb_e[2 * i], s.tb1);
    if (x.err) return;
    s.x2 = xx;
    t2 = x.ln(xx, rows, x.W(p + "norm2.weight"), x.W(p + "norm2.bias"));
    float* cat = x.alloc((size_t)rows * CFD_NMEM * D);           // torch.cat (cross_attention.py:629): column block j of [rows][5 D]
    cross_fwd_grouped(x, p, t2, mems, a.mem, att + (size_t)i * L * St, (long long)NL * L * St, cat, s.cross);
    if (x.err) return;
    if (i == NL - 1) break;                                      // nothing above the last cross-attention reaches the objective
    xx = x.linear(cat, CFD_NMEM * D, rows, CFD_NMEM * D, x.W(p + "att_fuser.weight"), x.W(p + "att_fuser.bias"), D, nullptr, 0, xx);
    xx = time_block_fwd(x, p + "time_block2", xx, tb_e[2 * i + 1], s.tb2);
    if (x.err) return;
    s.x4 = xx;
    t2 = x.ln(xx, rows, x.W(p + "norm3.weight"), x.W(p + "norm3.bias"));
    s.ffn_pre = x.linear(t2, D, rows, D, x.W(p + "linear1.weight"), x.W(p + "linear1.bias"), FF);
    xx = x.linear(s.ffn_pre, FF, rows, FF, x.W(p + "linear2.weight"), x.W(p + "linear2.bias"), D, nullptr, 0, xx, 2);   // x + W2 GELU(pre)
  }
  if (x.err) return;
  // ---- the objective and its gradient with respect to the nine maps
  const int W = a.last - 1;
  float* ws = x.alloc((size_t)B * (3 * (size_t)L * W + 3 * (size_t)a.nt_max));
  float* d_att = x.alloc((size_t)B * NL * L * St);
  if (!x.skip())
    hipLaunchKernelGGL(weg_focus_kernel<>, dim3((unsigned)B), dim3(256), 0, x.st, att, a.tok_off, a.tok_idx, B, NL, L, St, a.last, a.nt_max, a.k3[0],
                       a.k3[1], a.k3[2], ws, a.losses, a.max_att, d_att);
  // ---- reverse sweep
  float* g = x.alloc((size_t)rows * D);                          // gradient at the current layer's output
  float* dt2 = x.alloc((size_t)rows * D);
  bool have_g = false;
  for (int i = NL - 1; i >= 0; --i) {
    const std::string p = "decoder.layers." + std::to_string(i) + ".";
    const LayerSaved& s = sv[i];
    float* dcat = nullptr;
    if (have_g) {
      float* d1 = x.linear_bwd(g, D, rows, D, x.W(p + "linear2.weight"), FF);
      x.ew(EW_GELU_BWD, d1, s.ffn_pre, d1, rows * FF);
      float* d2 = x.linear_bwd(d1, FF, rows, FF, x.W(p + "linear1.weight"), D);
      x.ln_bwd(s.x4, x.W(p + "norm3.weight"), d2, g, rows, 1);
      time_block_bwd(x, s.tb2, g);
      dcat = x.linear_bwd(g, D, rows, D, x.W(p + "att_fuser.weight"), CFD_NMEM * D);
    }
    if (have_g) cross_bwd_grouped(x, s.cross, dcat, d_att + (size_t)i * L * St, (long long)NL * L * St, dt2);
    else mha_bwd(x, s.cross[2], nullptr, 0, d_att + (size_t)i * L * St, (long long)NL * L * St, false, dt2, 0);   // top layer: the objective only
    x.ln_bwd(s.x2, x.W(p + "norm2.weight"), dt2, g, rows, have_g ? 1 : 0);
    have_g = true;
    time_block_bwd(x, s.tb1, g);
    mha_bwd(x, s.self, g, D, nullptr, 0, true, dt2, 0);
    x.ln_bwd(s.x0, x.W(p + "norm1.weight"), dt2, g, rows, 1);
  }
  x.linear_bwd(g, D, rows, D, x.W("latent_embd.weight"), LAT, a.grad);
}

}  // namespace weg
