// libcfdenoise: word-excitation guidance -- the launch-by-launch float32 pieces (cfd_gemm_f32 ... cfd_weg_focus) and cfd_weg_eval: one evaluation of the word-excitation-guidance objective and its gradient, all launches enqueued from C++
// (float32 launch sequence: weg_eval.hpp; small problems on the row-tile kernels: rowtile_bwd.hpp, weg_rt.hpp).
#include "cfd_internal.hpp"
#include "grad.hpp"

#include "weg_eval.hpp"
#include "rowtile_bwd.hpp"
#include "weg_rt.hpp"

extern "C" int cfd_gemm_f32(cfd_handle c, int M, int N, int K, int nb1, int nb2, const cfd_mat* A, const cfd_mat* B, const cfd_mat* Cm,
                            const float* bias, float alpha, int accumulate, void* stream) {
  if (!c || !A || !B || !Cm || !A->p || !B->p || !Cm->p || M < 1 || N < 1 || K < 1 || nb1 < 1 || nb2 < 1) return fail(CFD_E_ARG, "bad argument");
  if ((long long)nb1 * nb2 > 65535) return fail(CFD_E_SHAPE, "cfd_gemm_f32: at most 65535 batch entries");
  HIPCHK(hipSetDevice(c->cfg.device));
  MatView a{A->p, A->rs, A->cs, A->b1, A->b2}, b{B->p, B->rs, B->cs, B->b1, B->b2};
  launch_gemm_f32((hipStream_t)stream, a, b, const_cast<float*>(Cm->p), Cm->rs, Cm->cs, Cm->b1, Cm->b2, M, N, K, nb1, nb2, bias, alpha, accumulate);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_softmax(cfd_handle c, float* scores, long long rows, int Lk, const uint8_t* key_padding_mask, long long rows_per_batch,
                           void* stream) {
  if (!c || !scores || rows < 1 || Lk < 1 || rows_per_batch < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(softmax_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, key_padding_mask, rows, Lk,
                     rows_per_batch);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_softmax_bwd(cfd_handle c, const float* p, float* dp, const float* extra, long long rows, int Lk, void* stream) {
  if (!c || !p || !dp || rows < 1 || Lk < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(softmax_bwd_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, extra, rows, Lk);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_layer_norm_bwd(cfd_handle c, const float* x, const float* gamma, const float* dy, float* dx, long long rows, int D, float eps,
                                  int accumulate, void* stream) {
  if (!c || !x || !gamma || !dy || !dx || rows < 1 || D < 1 || D > 2048) return fail(CFD_E_ARG, "bad argument (D <= 2048)");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(layernorm_bwd_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, dx, rows, D, eps,
                     accumulate, (const float*)nullptr, (const float*)nullptr);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_ew(cfd_handle c, int op, const float* a, const float* b, float* out, size_t numel, int D, int R1, long long s0, long long s1,
                      float alpha, void* stream) {
  if (!c || !a || !out || numel < 1 || op < 0 || op >= EW_NOPS) return fail(CFD_E_ARG, "bad argument");
  if (op >= EW_SILU_BWD && !b) return fail(CFD_E_ARG, "cfd_ew: this op needs the second operand");
  if (op >= EW_ADD_BCAST && (D < 1 || R1 < 1)) return fail(CFD_E_ARG, "cfd_ew: D and R1 must be positive");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(ew_f32_kernel<>, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, a, b, out, (long long)numel,
                     D > 0 ? D : 1, R1 > 0 ? R1 : 1, s0, s1, alpha);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_weg_focus(cfd_handle c, const float* att, int B, int NL, int L, int S, const int32_t* tok_off, const int32_t* tok_idx, int last,
                             int nt_max, const float kernel3[3], float* workspace, float* losses, float* max_att, float* d_att, void* stream) {
  if (!c || !att || !tok_off || !tok_idx || !kernel3 || !workspace || !losses || !max_att || !d_att || B < 1 || NL < 1 || nt_max < 1)
    return fail(CFD_E_ARG, "bad argument");
  // F.pad(..., mode='reflect') with pad 1 needs at least 2 entries per axis (word_excitation_guidance.py:35)
  if (L < 2 || last - 1 < 2 || last > S) return fail(CFD_E_SHAPE, "text slice [1, %d) of %d keys / %d frames is too short for the 3x3 reflect-padded smoothing", last, S, L);
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(weg_focus_kernel<>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, att, tok_off, tok_idx, B, NL, L, S, last, nt_max,
                     kernel3[0], kernel3[1], kernel3[2], workspace, losses, max_att, d_att);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_weg_eval(cfd_handle c, const cfd_weg_args* a, float* losses, float* max_att, float* grad, float* loss_host, void* stream) {
  if (!c || !a || !a->latents || !a->tok_off || !losses || !max_att || !grad) return fail(CFD_E_ARG, "null argument");
  if (!c->finalized) return fail(CFD_E_STATE, "weights not finalized");
  const int B = a->B, L = a->L, D = c->cfg.text_encoded_dim;
  if (B < 1 || L < 2) return fail(CFD_E_ARG, "bad batch / length");
  if (L % 2) return fail(CFD_E_SHAPE, "latent length %d is odd (reference: broadcasting error at position_encoding.py:160-161)", L);
  if (L / 2 > c->qpe_rows) return fail(CFD_E_SHAPE, "L/2 = %d exceeds the query PE buffer (%d rows)", L / 2, c->qpe_rows);
  if (a->timestep < 0 || a->timestep >= c->tsin_rows) return fail(CFD_E_ARG, "timestep %d outside the timestep table (%d rows)", a->timestep, c->tsin_rows);
  if (D > 2048) return fail(CFD_E_SHAPE, "model width above 2048");
  for (int j = 0; j < CFD_NMEM; ++j) {
    if (!a->mem[j].data || a->mem[j].S < 1) return fail(CFD_E_ARG, "memory %s missing", MEM_NAMES[j]);
    if (a->mem[j].U != B || a->mem[j].row_map) return fail(CFD_E_ARG, "cfd_weg_eval takes one memory per row (U == B, no row map)");
    if (a->mem[j].S > c->mpe_rows) return fail(CFD_E_SHAPE, "memory %s has %d tokens, the memory PE buffer %d rows", MEM_NAMES[j], a->mem[j].S, c->mpe_rows);
  }
  const int St = a->mem[2].S, n_tok = a->tok_off[B];
  if (a->tok_off[0] != 0 || n_tok < 0 || (n_tok > 0 && !a->tok_idx)) return fail(CFD_E_ARG, "bad focus-token table");
  // F.pad(..., mode='reflect') with pad 1 needs at least 2 entries per axis (word_excitation_guidance.py:35)
  if (a->last - 1 < 2 || a->last > St) return fail(CFD_E_SHAPE, "text slice [1, %d) of %d keys is too short for the 3x3 reflect-padded smoothing", a->last, St);
  int nt_max = 1;
  for (int b = 0; b < B; ++b) {
    if (a->tok_off[b + 1] < a->tok_off[b]) return fail(CFD_E_ARG, "bad focus-token table");
    nt_max = std::max(nt_max, a->tok_off[b + 1] - a->tok_off[b]);
  }
  for (int t = 0; t < n_tok; ++t)
    if (a->tok_idx[t] < 1 || a->tok_idx[t] > a->last - 1) return fail(CFD_E_ARG, "focus index %d is outside the text slice [1, %d)", a->tok_idx[t], a->last);
  HIPCHK(hipSetDevice(c->cfg.device));
  c->hint_now = c->hint_same_mem = false;
  CHK(settle_deferred_census(c));
  hipStream_t caller = (hipStream_t)stream;
  // the evaluation runs on the handle's own stream (capturable, and the one the sampling graph replays on: the two
  // serialise); it starts behind whatever the caller has queued on `stream`
  hipStream_t st = c->own_stream;
  HIPCHK(hipEventRecord(c->weg_ev, caller));
  HIPCHK(hipStreamWaitEvent(st, c->weg_ev, 0));
  std::vector<int32_t> tok(a->tok_off, a->tok_off + B + 1);
  tok.insert(tok.end(), a->tok_idx, a->tok_idx + n_tok);
  if (tok != c->weg_tok_host) {                       // focus-token tables to the device (the stream may still read the old copy)
    HIPCHK(hipStreamSynchronize(st));
    CHK(c->weg_tok.ensure((size_t)(B + 1 + std::max(1, n_tok)) * 4));
    HIPCHK(hipMemcpy(c->weg_tok.p, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));
    c->weg_tok_host = tok;
    ++c->weg_tok_version;
  }
  // staging: [latents | timestep row | losses | max_att | grad]
  const size_t n_lat = (size_t)B * L * CFD_LAT, n_max = (size_t)std::max(1, n_tok);
  const size_t o_lat = 0, o_trow = o_lat + n_lat, o_loss = o_trow + (size_t)D, o_max = o_loss + (size_t)((B + 63) / 64 * 64),
               o_grad = o_max + (n_max + 63) / 64 * 64, n_io = o_grad + n_lat;
  if (n_io * 4 > c->weg_io.bytes) {
    HIPCHK(hipStreamSynchronize(st));
    CHK(c->weg_io.ensure(n_io * 4));
  }
  float* io = c->weg_io.as<float>();
  HIPCHK(hipMemcpyAsync(io + o_lat, a->latents, n_lat * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(io + o_trow, c->tsin.as<float>() + (size_t)a->timestep * D, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
  weg::Args wa{io + o_lat, io + o_trow, a->mem, c->weg_tok.as<int32_t>(), c->weg_tok.as<int32_t>() + B + 1,
               a->last, nt_max, {a->kernel3[0], a->kernel3[1], a->kernel3[2]}, io + o_loss, io + o_max, io + o_grad};
  // what the memory-side / time-only part of an evaluation depends on: with args->reuse_memory_side the caller states that the
  // memories' CONTENTS are unchanged too (a refinement loop at one timestep), and those launches are skipped
  // What the memory-side / time-only part of an evaluation depends on.  With args->reuse_memory_side the caller states that the
  // memories' CONTENTS are unchanged too, and those launches are skipped: 1 = same timestep as well (a refinement loop at one
  // timestep), 2 = the timestep may differ (the guided sampling loop: one evaluation per iteration, same conditioning).  The
  // row-tile path serves 2 from tables over ALL timesteps, built at the first such call (row t = timestep t, one launch per
  // evaluation copies the row); the float32 launch sequence treats 2 with a new timestep as 0.
  const bool use_rt = wegrt::eligible(c, a);
  std::vector<long long> sig = {B, L};
  for (int j = 0; j < CFD_NMEM; ++j) {
    sig.push_back(a->mem[j].S);
    sig.push_back((long long)(size_t)a->mem[j].data);
    sig.push_back((long long)(size_t)a->mem[j].key_padding_mask);   // (wegrt::prepare rebuilds the problem when a mask pointer changes: no reuse then)
  }
  weg::Ctx x{c, st, true, reinterpret_cast<char*>(256), 0, B, L, D, D, CFD_OK, std::string(), 0};
  // small problems (the product shape) run on the row-tile kernels, everything else on the float32 launch sequence of weg_eval.hpp
  const void* arena = nullptr;
  bool reuse = false;
  if (use_rt) {
    const bool had_full = c->wrt.T > 1;
    // tables over all timesteps stay while the caller keeps stating that the conditioning is unchanged
    const int T = a->reuse_memory_side == 2 || (a->reuse_memory_side == 1 && had_full) ? c->tsin_rows : 1;
    CHK(wegrt::prepare(c, a, T, st));
    arena = c->weg_rt_ws.p;
    sig.push_back((long long)(size_t)arena);
    sig.push_back(-(long long)T);
    if (T == 1) sig.push_back(a->timestep);
    reuse = a->reuse_memory_side != 0 && sig == c->weg_sig;
    c->wrt.T = T;
    c->weg_t_host = a->timestep;                     // in front of the launch sequence, outside any captured graph
    c->weg_dstep_host = T > 1 ? a->timestep : 0;
    if (T == 1) HIPCHK(hipMemcpyAsync(c->wk[1].trows.p, &c->weg_t_host, 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->wk[1].d_step.p, &c->weg_dstep_host, 4, hipMemcpyHostToDevice, st));
  } else {
    sig.push_back(a->timestep);
    weg::run(x, wa);                                 // sizing pass
    if (x.err) return fail(x.err, "missing tensor '%s' (state-dict key denoiser.%s)", x.missing.c_str(), x.missing.c_str());
    if (x.off > c->weg_ws.bytes) HIPCHK(hipStreamSynchronize(st));
    CHK(c->weg_ws.ensure(x.off));
    arena = c->weg_ws.p;
    sig.push_back((long long)(size_t)arena);
    sig.push_back((long long)x.off);
    reuse = a->reuse_memory_side != 0 && sig == c->weg_sig;
  }
  x.dry = false;
  x.base = c->weg_ws.as<char>();
  x.off = 0;
  x.launches = 0;
  x.reuse = reuse;
  c->weg_sig.clear();
  const wegrt::EvalArgs ea{io + o_lat, c->weg_tok.as<int32_t>(), c->weg_tok.as<int32_t>() + B + 1, a->last, nt_max,
                           {a->kernel3[0], a->kernel3[1], a->kernel3[2]}, io + o_loss, io + o_max, io + o_grad};
  auto run_eval = [&]() -> int {                     // the evaluation's launches (this is what a graph captures)
    if (!use_rt) { weg::run(x, wa); return CFD_OK; }
    const int r = wegrt::enqueue(c, st, !x.reuse, ea);
    x.launches = c->wrt.launches;
    return r;
  };
  // everything the launch sequence and its (by-value) kernel arguments depend on, the timestep excepted (its row is staged)
  std::vector<long long> key = {B, L, a->last, nt_max, c->weg_tok_version, (long long)(size_t)c->weg_tok.p, (long long)(size_t)io, (long long)n_io,
                                (long long)(size_t)arena, (long long)x.reuse, (long long)use_rt, (long long)(use_rt ? c->wrt.T : 0)};
  for (int j = 0; j < CFD_NMEM; ++j) {
    key.push_back(a->mem[j].S);
    key.push_back((long long)(size_t)a->mem[j].data);
    key.push_back((long long)(size_t)a->mem[j].key_padding_mask);
  }
  for (int k = 0; k < 3; ++k) { long long bits = 0; memcpy(&bits, &a->kernel3[k], 4); key.push_back(bits); }
  auto& wg = c->weg_graph[x.reuse ? 1 : 0];
  if (wg.key != key) {
    if (wg.exec) { (void)hipGraphExecDestroy(wg.exec); wg.exec = nullptr; }
    if (wg.graph) { (void)hipGraphDestroy(wg.graph); wg.graph = nullptr; }
    wg.key = key;
    wg.uses = 0;
  }
  if (c->weg_graph_on && wg.exec) {
    HIPCHK(hipGraphLaunch(wg.exec, st));
    x.launches = c->weg_launches;
  } else if (c->weg_graph_on && wg.uses >= 1) {       // second use of this key: capture, instantiate, launch
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rr = run_eval();
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rr != CFD_OK) { if (g) (void)hipGraphDestroy(g); return rr; }
    if (e != hipSuccess) return fail(CFD_E_HIP, "capturing the WEG evaluation failed: %s", hipGetErrorString(e));
    wg.graph = g;
    HIPCHK(hipGraphInstantiate(&wg.exec, wg.graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphLaunch(wg.exec, st));
  } else {
    CHK(run_eval());
  }
  ++wg.uses;
  HIPCHK(hipGetLastError());
  c->weg_launches = x.launches;
  c->weg_sig = sig;
  HIPCHK(hipMemcpyAsync(losses, io + o_loss, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(max_att, io + o_max, n_max * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(grad, io + o_grad, n_lat * 4, hipMemcpyDeviceToDevice, st));
  if (loss_host) {                                   // torch.mean(losses) over the batch (word_excitation_guidance.py:80)
    std::vector<float> l(B);
    HIPCHK(hipMemcpyAsync(l.data(), io + o_loss, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    CHK(check_saturation(c, "cfd_weg_eval (latents, memories / their projections)"));   // (without loss_host: read by the next call that waits on this handle)
    float sum = 0.f;
    for (int b = 0; b < B; ++b) sum += l[b];
    *loss_host = sum / (float)B;
  } else {                                           // the caller's stream continues behind the evaluation
    c->census_pending = true;                        // (read by the handle's next entry point: settle_deferred_census)
    HIPCHK(hipEventRecord(c->weg_ev, st));
    HIPCHK(hipStreamWaitEvent(caller, c->weg_ev, 0));
  }
  return CFD_OK;
}

